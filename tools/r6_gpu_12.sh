#!/bin/bash
# round 6, call 12: the G2 accumulation on a stream of its own (beside the G1 accumulations) against the one accumulation stream; plain loads again
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_12; mkdir -p $O
export ZKR_EXP_MERGE=0
for r in 1 2 3; do
  ( export ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so; python3 tools/rate_only.py 20 40 pipelined_r5 )
  ( python3 tools/rate_only.py 20 40 pipelined_cur )
  ( export ZKR_EXP_B2_AUX=1; python3 tools/rate_only.py 20 40 pipelined_b2_aux )
done 2>&1 | grep '^\[' | tee $O/b2_aux.txt
