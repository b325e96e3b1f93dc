# Round-6 measurement set on one MI355X (run through gpurun):  bash tools/profile_round6.sh <part> <tag>   -> gpurun_out/<tag>/
#   part a: the default bench as the driver runs it (wall time recorded) + the synchronous leg
#   part c: bench lines of the other BASELINE sizes (2^22 with its 8-shard legs, 2^24 rollup-shaped and dense) and the in-process two-replica rehearsal
#   part b: ONE set for DESIGN.md section 5: kernel stats of the benchmarked command (rocprofv3 --kernel-trace --stats), queue occupancy,
#           steady-state timeline of one proof period, isolated kernel durations (ZKR_SERIAL=1), HBM traffic (FETCH_SIZE / WRITE_SIZE in
#           separate --pmc passes, calibrated per access pattern: profiles/summarize_pmc.py), the VALU census (SQ counters, one pass)
# rocprofv3 is always given the program itself after `--` (python3 ...), and --pmc passes carry --kernel-trace only.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
PART=${1:-a}; O=gpurun_out/${2:-r6p}; mkdir -p $O
LIGHT="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --no-2-22 --no-withdraw --shards 0"
if [ $PART = a ]; then
  S=$(date +%s); python3 bench.py > $O/bench_default.json 2>$O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - S )) s wall" > $O/bench_default_wall.txt
  python3 bench.py --no-pipeline $LIGHT > $O/bench_sync.json 2>/dev/null
  cat $O/bench_default_wall.txt; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; c=d['config']; cb=d.get('cpu_baseline') or {}
print(d['value'], 'proofs/s', d['ms_per_step'], 'ms | frac', r['frac'], 'traffic_over_algorithmic', r.get('traffic_over_algorithmic'), 'valu_busy_step', r.get('valu_busy_step'), '| host buffer sync', c.get('host_buffer_sync_proofs_per_s'), 'sync ms', c.get('sync_latency_ms'), '| 2^22', c.get('rate_2_22_proofs_per_s'), '| snarkjs-style x', cb.get('speedup_vs_snarkjs_style'))"
fi
if [ $PART = b ]; then
  rocprofv3 --kernel-trace --stats -d $O/trace -- python3 bench.py --steps 20 --warmup 3 $LIGHT > $O/bench_traced.json 2>$O/trace.err
  DB=$(find $O/trace -name "*.db" | head -1)
  python3 profiles/summarize_rocpd.py $DB 0 > $O/kernel_stats.md
  python3 profiles/occupancy_timeline.py $DB > $O/queue_occupancy.txt
  python3 profiles/steady_timeline.py $DB > $O/steady_timeline.md; rm -rf $O/trace
  ZKR_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/strace -- python3 bench.py --steps 4 --warmup 1 --no-pipeline $LIGHT > $O/bench_serial.json 2>$O/strace.err
  python3 profiles/summarize_rocpd.py $(find $O/strace -name "*.db" | head -1) 0 > $O/serial_kernel_stats.md; rm -rf $O/strace
  ZKR_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -- python3 bench.py --steps 2 --warmup 1 --no-pipeline $LIGHT > /dev/null 2>$O/pf.err
  ZKR_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -- python3 bench.py --steps 2 --warmup 1 --no-pipeline $LIGHT > /dev/null 2>$O/pw.err
  python3 profiles/summarize_pmc.py $(find $O/pf -name "*.db" | head -1) $(find $O/pw -name "*.db" | head -1) $O/pmc_traffic.json 20 6 isolated > $O/pmc_traffic.md; rm -rf $O/pf $O/pw
  SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES"
  ZKR_SERIAL=1 timeout 900 rocprofv3 --kernel-trace --pmc $SQ GRBM_GUI_ACTIVE -d $O/cs -- python3 bench.py --steps 2 --warmup 1 --no-pipeline $LIGHT > /dev/null 2>$O/cs.err
  STEP=$(python3 -c "import json; print(json.load(open('$O/bench_traced.json'))['ms_per_step'])")
  python3 profiles/summarize_census.py $(find $O/cs -name "*.db" | head -1) auto $STEP 2.27 $O/valu_census.json 20 6 > $O/valu_census.md; rm -rf $O/cs
  head -14 $O/kernel_stats.md | cut -c1-160; head -12 $O/pmc_traffic.md; tail -4 $O/valu_census.md
fi
if [ $PART = c ]; then
  python3 bench.py --log-m 22 --steps 20 $LIGHT --shards 8 > $O/bench_2_22.json 2>/dev/null
  python3 bench.py --log-m 24 --steps 6 --warmup 1 $LIGHT --shards 8 > $O/bench_2_24_rollup.json 2>/dev/null
  python3 bench.py --log-m 24 --shape dense --steps 6 --warmup 1 $LIGHT --shards 8 > $O/bench_2_24_dense.json 2>/dev/null
  python3 bench.py --gpus 2 --inproc --devices 0,0 --steps 20 --warmup 3 > $O/bench_inproc_0_0.json 2>/dev/null
  for f in bench_2_22 bench_2_24_rollup bench_2_24_dense bench_inproc_0_0; do python3 -c "
import json; d=json.load(open('$O/$f.json')); s=d.get('intra_proof_sharding') or {}
print('$f', round(d['value'],2), 'proofs/s', round(d['ms_per_step'],2), 'ms', '| shards:', s.get('parts'), 'whole', s.get('whole_key_sync_proof_ms') and round(s['whole_key_sync_proof_ms'],2), 'slowest', s.get('slowest_shard_ms') and round(s['slowest_shard_ms'],2), 'form', s.get('form'))"; done
fi
