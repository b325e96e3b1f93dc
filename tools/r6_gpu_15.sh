#!/bin/bash
# round 6, call 15: robustness of the last tree -- concurrent callers over a key, a replica and four shards for two minutes at 2^16 and
# one at 2^20 (every proof verified, device memory flat), the key lifecycle loop, the multi-GPU preflight rehearsed on devices 0,0
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_15; mkdir -p $O
python3 tests/soak.py 16 120 > $O/soak_2_16.txt 2>&1; tail -4 $O/soak_2_16.txt
python3 tests/soak.py 20 60 > $O/soak_2_20.txt 2>&1; tail -4 $O/soak_2_20.txt
python3 tools/key_lifecycle.py 16 6 > $O/key_lifecycle.txt 2>&1; tail -3 $O/key_lifecycle.txt
python3 tools/multi_gpu_preflight.py --devices 0,0 > $O/preflight_0_0.jsonl 2>&1; tail -3 $O/preflight_0_0.jsonl | cut -c1-300
