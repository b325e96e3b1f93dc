#!/bin/bash
# round 6, call 11: table points gathered with four / eight ALIGNED 16-byte non-temporal loads (the compiler had regrouped them into
# six / twelve 4-byte-aligned ones) -- parity, isolated launch times, pipelined rates against round 5's library
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_11; mkdir -p $O
python -m pytest tests/test_gpu_stages.py -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log; tail -3 $O/tests_gpu.log
export ZKR_EXP_MERGE=0
for r in 1 2 3; do
  ( export ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so; ZKR_SERIAL=1 python3 tools/rate_only.py 20 12 serial_r5; python3 tools/rate_only.py 20 40 pipelined_r5 )
  ( ZKR_SERIAL=1 python3 tools/rate_only.py 20 12 serial_aligned; python3 tools/rate_only.py 20 40 pipelined_aligned )
  ( export ZKR_EXP_FEED=2; ZKR_SERIAL=1 python3 tools/rate_only.py 20 12 serial_aligned_regs2; python3 tools/rate_only.py 20 40 pipelined_aligned_regs2 )
  ( export ZKR_EXP_FEED=0; ZKR_SERIAL=1 python3 tools/rate_only.py 20 12 serial_aligned_g2lds; python3 tools/rate_only.py 20 40 pipelined_aligned_g2lds )
done 2>&1 | grep '^\[' | tee $O/aligned_loads.txt
