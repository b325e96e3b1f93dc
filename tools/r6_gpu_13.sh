#!/bin/bash
# round 6, call 13: the cleaned tree -- whole GPU suite, smoke, then same-box A/B of the headline against round 5's library
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_13; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log; tail -4 $O/tests_gpu.log
python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
for r in 1 2 3; do
  ( export ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so; python3 tools/rate_only.py 20 40 pipelined_r5 )
  ( python3 tools/rate_only.py 20 40 pipelined_r6 )
done 2>&1 | grep '^\[' | tee $O/ab_r5_r6.txt
