cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r11; mkdir -p $O

python bench.py > $O/bench.json 2>$O/bench.err
python bench.py --no-pipeline --no-cpu-baseline --no-tx-circuit > $O/bench_sync.json 2>/dev/null
python bench.py --log-m 22 --steps 20 --no-cpu-baseline --no-tx-circuit > $O/bench_2_22.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-tx-circuit > $O/bench_traced.json 2>$O/trace.err
python3 profiles/summarize_rocpd.py $(find $O/trace -name "*.db" | head -1) 0 > $O/kernel_stats.md; rm -rf $O/trace
ZKR_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/strace -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pipeline --no-tx-circuit > $O/bench_serial.json 2>$O/strace.err
python3 profiles/summarize_rocpd.py $(find $O/strace -name "*.db" | head -1) 0 > $O/serial_kernel_stats.md; rm -rf $O/strace
ZKR_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-tx-circuit > /dev/null 2>$O/pf.err
ZKR_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-tx-circuit > /dev/null 2>$O/pw.err
python3 profiles/summarize_pmc.py $(find $O/pf -name "*.db" | head -1) $(find $O/pw -name "*.db" | head -1) $O/pmc_traffic.json 20 > $O/pmc_traffic.md; rm -rf $O/pf $O/pw
tail -c 300 $O/bench.json; ls -la $O
