# Round-2 measurement set on one MI355X (run through gpurun): bench lines of every BASELINE size, kernel stats of the
# benchmarked command (rocprofv3 --kernel-trace --stats), isolated kernel durations (ZKR_SERIAL=1), HBM traffic (PMC
# FETCH_SIZE / WRITE_SIZE in separate passes).  Usage: bash tools/profile_round2.sh <tag>   -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/${1:-r2p}; mkdir -p $O
python bench.py > $O/bench.json 2>$O/bench.err
python bench.py --no-pipeline --no-cpu-baseline --no-tx-circuit --no-bcast-modes --shards 0 > $O/bench_sync.json 2>/dev/null
python bench.py --log-m 22 --steps 20 --no-cpu-baseline --no-tx-circuit --no-bcast-modes --shards 0 > $O/bench_2_22.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-tx-circuit --no-bcast-modes --shards 0 > $O/bench_traced.json 2>$O/trace.err
python3 profiles/summarize_rocpd.py $(find $O/trace -name "*.db" | head -1) 0 > $O/kernel_stats.md; rm -rf $O/trace
ZKR_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/strace -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pipeline --no-tx-circuit --no-bcast-modes --shards 0 > $O/bench_serial.json 2>$O/strace.err
python3 profiles/summarize_rocpd.py $(find $O/strace -name "*.db" | head -1) 0 > $O/serial_kernel_stats.md; rm -rf $O/strace
ZKR_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-tx-circuit --no-bcast-modes --shards 0 > /dev/null 2>$O/pf.err
ZKR_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-tx-circuit --no-bcast-modes --shards 0 > /dev/null 2>$O/pw.err
python3 profiles/summarize_pmc.py $(find $O/pf -name "*.db" | head -1) $(find $O/pw -name "*.db" | head -1) $O/pmc_traffic.json 20 2 > $O/pmc_traffic.md; rm -rf $O/pf $O/pw
python bench.py --log-m 24 --steps 6 --warmup 1 --no-cpu-baseline --no-tx-circuit --no-bcast-modes --shards 0 > $O/bench_2_24_rollup.json 2>/dev/null
python bench.py --log-m 24 --shape dense --steps 6 --warmup 1 --no-cpu-baseline --no-tx-circuit --no-bcast-modes --shards 0 > $O/bench_2_24_dense.json 2>/dev/null
for f in bench bench_sync bench_2_22 bench_traced bench_2_24_rollup bench_2_24_dense; do python3 -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['value'],2), 'proofs/s', round(d['ms_per_step'],2), 'ms', d.get('tx_circuit',{}).get('proofs_per_s'))"; done
