#!/bin/bash
# round 6, call 3: why the merged accumulation launch is SLOWER (143.4 against 151.2 proofs/s, profiles/r6_02_ab_merged_launch.txt):
# kernel traces of both libraries' pipelined schedules (steady-state timeline of one proof period, per-kernel in-flight durations)
# and the groupings in between (ZKR_EXP_MERGE, a temporary knob of this tree)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_03; mkdir -p $O
LIGHT="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --no-2-22 --no-withdraw --shards 0"
run() { local name=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    python3 bench.py --steps 30 --warmup 5 $LIGHT 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['config']; st=d['stage_ms_per_proof']
print('%-10s %.2f proofs/s  %.3f ms  sync %.2f ms  sclk %.0f  power %.0f W | acc_g1 %.2f acc_g2 %.2f ntt %.2f sort %.2f big %.2f reduce %.2f total %.2f' % ('$name', d['value'], d['ms_per_step'], b.get('sync_latency_ms') or 0, d['device_state_during_timed_region']['sclk_mhz_mean'], d['device_state_during_timed_region'].get('power_w_mean') or 0, st['msm_accum_g1'], st['msm_accum_g2'], st['ntt'], st['msm_sort'], st['msm_big'], st['msm_reduce'], st['total']))" ) | tee -a $O/ab_groupings.txt
}
run r5 ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so
for m in 0 1 2 3; do run merge$m ZKR_EXP_MERGE=$m; done
run r5 ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so
for m in 0 3; do run merge$m ZKR_EXP_MERGE=$m; done
trace() { local name=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    rocprofv3 --kernel-trace --stats -d $O/t_$name -- python3 bench.py --steps 12 --warmup 3 $LIGHT > $O/bench_traced_$name.json 2>$O/trace_$name.err )
  DB=$(find $O/t_$name -name "*.db" | head -1)
  python3 profiles/summarize_rocpd.py $DB 0 > $O/kernel_stats_$name.md
  python3 profiles/occupancy_timeline.py $DB > $O/queue_occupancy_$name.txt
  python3 profiles/steady_timeline.py $DB > $O/steady_timeline_$name.md
  rm -rf $O/t_$name
}
trace r5 ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so
trace merge3 ZKR_EXP_MERGE=3
trace merge0 ZKR_EXP_MERGE=0
head -12 $O/queue_occupancy_r5.txt; head -12 $O/queue_occupancy_merge3.txt
