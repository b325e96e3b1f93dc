#!/bin/bash
# round 5: the round's last tree -- whole GPU suite, smoke, default bench as the driver runs it (wall time recorded)
O=gpurun_out/r5_09; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log
python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
S=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - S )) s wall" > $O/bench_default_wall.txt
tail -3 $O/tests_gpu.log; tail -2 $O/smoke.txt; cat $O/bench_default_wall.txt
