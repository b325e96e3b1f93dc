#!/bin/bash
# round 6, call 17: the multiplier microbenchmark sustained for seconds (is 181 G Fq-mul/s a burst figure?), then the pipelined
# bench on the same box for the same unit
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_17; mkdir -p $O
python3 tools/sustained_mul.py 12 2>&1 | grep -v amdgpu.ids | tee $O/sustained_mul.txt
python3 tools/rate_only.py 20 60 pipelined 2>&1 | grep '^\[' | tee -a $O/sustained_mul.txt
python3 tools/sustained_mul.py 4 2>&1 | grep -v amdgpu.ids | tee -a $O/sustained_mul.txt
