cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r2t; mkdir -p $O
rocprofv3 --kernel-trace -d $O/trace -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-tx-circuit --no-bcast-modes --shards 0 > $O/bench_traced.json 2>$O/trace.err
python3 profiles/occupancy_timeline.py $(find $O/trace -name "*.db" | head -1) > $O/occupancy.txt; python3 profiles/timeline.py $(find $O/trace -name "*.db" | head -1) 3 > $O/timeline_one_proof.txt; rm -rf $O/trace; cat $O/occupancy.txt
