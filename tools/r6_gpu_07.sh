#!/bin/bash
# round 6, call 7: a BOUND before building deeper point prefetch -- a measurement-only build whose gathers all land in the first 1 MB of
# their table (L2-resident: garbage sums, the memory side of the accumulation removed): isolated launch times and the pipelined rate
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_07; mkdir -p $O
for r in 1 2; do
for v in cur fakegather; do
  ( [ $v != cur ] && export ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_$v.so; export ZKR_EXP_MERGE=0
    ZKR_SERIAL=1 python3 tools/rate_only.py 20 12 serial_$v; python3 tools/rate_only.py 20 40 pipelined_$v )
done
done 2>&1 | grep '^\[' | tee $O/fake_gather_bound.txt
