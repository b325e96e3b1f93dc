# Round-5 measurement set (the round-4 script with the new side legs switched off in LIGHT) on one MI355X (run through gpurun):  bash tools/profile_round5.sh <part> <tag>   -> gpurun_out/<tag>/
#   part a: bench lines of every BASELINE size (2^20 synchronous, 2^22, 2^24 rollup-shaped and dense), the intra-proof sharding
#           legs at 2^22 / 2^24 (8 shards side by side: projected one-shard-per-GPU latency) and the in-process two-replica rehearsal
#   part b: ONE set for DESIGN.md section 5: kernel stats of the benchmarked command (rocprofv3 --kernel-trace --stats), queue
#           occupancy, isolated kernel durations (ZKR_SERIAL=1), HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) for
#           BOTH schedules (isolated and pipelined), SQ_INSTS_VALU / SQ_WAVES per kernel
#   part c: the counter set BASELINE configs[2] asks for at 2^22 (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU, SQ_WAVES, LDS bank conflicts), isolated kernels
#   part d: timeline of one synchronous proof of the tx circuit (2^17) -- launches per proof -- and of one 2^20 proof
# rocprofv3 is always given the program itself after `--` (python3 ...), and --pmc passes carry --kernel-trace only.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
PART=${1:-a}; O=gpurun_out/${2:-r5p}; mkdir -p $O
LIGHT="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --no-2-22 --no-withdraw --shards 0"
if [ $PART = a ]; then
  python3 bench.py --no-pipeline $LIGHT > $O/bench_sync.json 2>/dev/null
  python3 bench.py --log-m 22 --steps 20 $LIGHT --shards 8 > $O/bench_2_22.json 2>/dev/null
  python3 bench.py --log-m 24 --steps 6 --warmup 1 $LIGHT --shards 8 > $O/bench_2_24_rollup.json 2>/dev/null
  python3 bench.py --log-m 24 --shape dense --steps 6 --warmup 1 $LIGHT --shards 8 > $O/bench_2_24_dense.json 2>/dev/null
  python3 bench.py --gpus 2 --inproc --devices 0,0 --steps 20 --warmup 3 > $O/bench_inproc_0_0.json 2>/dev/null
  S=$(date +%s); python3 bench.py --with-2-24-dense > $O/bench_default_with_2_24_dense.json 2>$O/bench_default.err; echo "default bench + 2^24 dense leg: $(( $(date +%s) - S )) s wall" > $O/bench_default_wall.txt
  for f in bench_sync bench_2_22 bench_2_24_rollup bench_2_24_dense bench_inproc_0_0; do python3 -c "
import json; d=json.load(open('$O/$f.json')); s=d.get('intra_proof_sharding') or {}
print('$f', round(d['value'],2), 'proofs/s', round(d['ms_per_step'],2), 'ms', '| shards:', s.get('parts'), 'whole', s.get('whole_key_sync_proof_ms') and round(s['whole_key_sync_proof_ms'],2), 'slowest', s.get('slowest_shard_ms') and round(s['slowest_shard_ms'],2))"; done
fi
if [ $PART = b ]; then
  rocprofv3 --kernel-trace --stats -d $O/trace -- python3 bench.py --steps 20 --warmup 3 $LIGHT > $O/bench_traced.json 2>$O/trace.err
  python3 profiles/summarize_rocpd.py $(find $O/trace -name "*.db" | head -1) 0 > $O/kernel_stats.md
  python3 profiles/occupancy_timeline.py $(find $O/trace -name "*.db" | head -1) > $O/queue_occupancy.txt; rm -rf $O/trace
  ZKR_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/strace -- python3 bench.py --steps 4 --warmup 1 --no-pipeline $LIGHT > $O/bench_serial.json 2>$O/strace.err
  python3 profiles/summarize_rocpd.py $(find $O/strace -name "*.db" | head -1) 0 > $O/serial_kernel_stats.md; rm -rf $O/strace
  ZKR_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -- python3 bench.py --steps 2 --warmup 1 --no-pipeline $LIGHT > /dev/null 2>$O/pf.err
  ZKR_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -- python3 bench.py --steps 2 --warmup 1 --no-pipeline $LIGHT > /dev/null 2>$O/pw.err
  python3 profiles/summarize_pmc.py $(find $O/pf -name "*.db" | head -1) $(find $O/pw -name "*.db" | head -1) $O/pmc_traffic.json 20 4 isolated > $O/pmc_traffic.md; rm -rf $O/pf $O/pw
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -- python3 bench.py --steps 8 --warmup 2 $LIGHT > /dev/null 2>$O/pfp.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -- python3 bench.py --steps 8 --warmup 2 $LIGHT > /dev/null 2>$O/pwp.err
  python3 profiles/summarize_pmc.py $(find $O/pf -name "*.db" | head -1) $(find $O/pw -name "*.db" | head -1) $O/pmc_traffic_pipelined.json 20 4 pipelined > $O/pmc_traffic_pipelined.md; rm -rf $O/pf $O/pw
  for c in SQ_INSTS_VALU SQ_WAVES; do
    ZKR_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/$c -- python3 bench.py --steps 2 --warmup 1 --no-pipeline $LIGHT > /dev/null 2>$O/$c.err
    DB=$(find $O/$c -name "*.db" | head -1)
    if [ -n "$DB" ]; then python3 profiles/summarize_counter.py $DB $c > $O/$c.md; else tail -3 $O/$c.err; fi
    rm -rf $O/$c
  done
  head -14 $O/kernel_stats.md | cut -c1-160; head -12 $O/pmc_traffic.md; head -12 $O/pmc_traffic_pipelined.md
fi
if [ $PART = c ]; then
  for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT; do
    ZKR_SERIAL=1 timeout 900 rocprofv3 --kernel-trace --pmc $c -d $O/$c -- python3 bench.py --log-m 22 --steps 2 --warmup 1 --no-pipeline $LIGHT > /dev/null 2>$O/$c.err
    DB=$(find $O/$c -name "*.db" | head -1)
    if [ -n "$DB" ]; then python3 profiles/summarize_counter.py $DB $c > $O/c22_$c.md; else tail -3 $O/$c.err; fi
    rm -rf $O/$c
  done
  ZKR_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/st22 -- python3 bench.py --log-m 22 --steps 3 --warmup 1 --no-pipeline $LIGHT > $O/bench_2_22_serial.json 2>$O/st22.err
  python3 profiles/summarize_rocpd.py $(find $O/st22 -name "*.db" | head -1) 0 > $O/serial_kernel_stats_2_22.md; rm -rf $O/st22
  head -12 $O/c22_FETCH_SIZE.md; head -12 $O/c22_SQ_LDS_BANK_CONFLICT.md
fi
if [ $PART = d ]; then
  rocprofv3 --kernel-trace -d $O/ttx -- python3 tools/tx_single.py 12 > $O/tx_single.txt 2>$O/ttx.err
  python3 profiles/timeline.py $(find $O/ttx -name "*.db" | head -1) 8 > $O/timeline_one_tx_proof.txt; rm -rf $O/ttx
  rocprofv3 --kernel-trace -d $O/t20 -- python3 tools/sync_single.py 20 8 > $O/sync_single.txt 2>$O/t20.err
  python3 profiles/timeline.py $(find $O/t20 -name "*.db" | head -1) 3 > $O/timeline_one_2_20_proof.txt; rm -rf $O/t20
  cat $O/tx_single.txt; head -80 $O/timeline_one_tx_proof.txt
fi
