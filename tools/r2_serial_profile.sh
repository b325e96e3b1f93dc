# isolated kernel durations of the current build: one stream, one proof at a time (ZKR_SERIAL=1), rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/${1:-r2serial}; mkdir -p $O
ZKR_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/strace -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pipeline --no-tx-circuit --no-bcast-modes --shards 0 > $O/bench_serial.json 2>$O/strace.err
python3 profiles/summarize_rocpd.py $(find $O/strace -name "*.db" | head -1) 0 > $O/serial_kernel_stats.md; rm -rf $O/strace
head -30 $O/serial_kernel_stats.md | cut -c1-200
