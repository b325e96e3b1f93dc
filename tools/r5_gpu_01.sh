#!/bin/bash
# round 5, first GPU call: the new / changed GPU tests, the default bench (timed: the driver's budget is a few minutes), the upload bound
O=gpurun_out/r5_01; mkdir -p $O
python -m pytest tests/test_gpu_shard.py tests/test_gpu_multi.py -m gpu -x -q > $O/tests_shard.log 2>&1; echo "rc=$?" >> $O/tests_shard.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "bench or preflight" > $O/tests_bench.log 2>&1; echo "rc=$?" >> $O/tests_bench.log
S=$(date +%s.%N); python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$? wall_s=$(echo "$(date +%s.%N) - $S" | bc)" > $O/bench_default.time
python tools/sync_single.py 20 16 > $O/sync_single.txt 2>&1
python tools/multi_gpu_preflight.py --devices 0,0 > $O/preflight.jsonl 2> $O/preflight.err
tail -3 $O/tests_shard.log $O/tests_bench.log; cat $O/bench_default.time; tail -2 $O/sync_single.txt; tail -1 $O/preflight.jsonl
