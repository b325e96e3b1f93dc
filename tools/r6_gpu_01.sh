#!/bin/bash
# round 6, call 1: baseline rate on this box + the VALU census VERDICT r5 asks for (next 1a).
# One --pmc pass carries the SQ counters together (8 SQ slots on gfx950) + GRBM_GUI_ACTIVE; if the pass is refused it is split.
# Isolated kernels (ZKR_SERIAL=1) give each kernel's own VALU occupancy; the pipelined pass shows what dispatch-mode counters see
# of the benchmarked schedule (rocprofv3 serialises counted dispatches: check its kernel durations against the un-profiled ones).
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_01; mkdir -p $O
LIGHT="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --no-2-22 --no-withdraw --shards 0"
python3 bench.py --steps 40 --warmup 5 $LIGHT > $O/bench_light.json 2>$O/bench_light.err
python3 -c "
import json; d=json.load(open('$O/bench_light.json')); print('baseline', d['value'], 'proofs/s', d['ms_per_step'], 'ms; sclk', d['device_state_during_timed_region']['sclk_mhz_mean'])" | tee $O/baseline.txt
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES"
ZKR_SERIAL=1 timeout 900 rocprofv3 --kernel-trace --pmc $SQ GRBM_GUI_ACTIVE -d $O/cs -- python3 bench.py --steps 2 --warmup 1 --no-pipeline $LIGHT > /dev/null 2>$O/cs.err
DB=$(find $O/cs -name "*.db" | head -1)
if [ -z "$DB" ]; then
  tail -5 $O/cs.err
  ZKR_SERIAL=1 timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/cs -- python3 bench.py --steps 2 --warmup 1 --no-pipeline $LIGHT > /dev/null 2>$O/cs2.err
  DB=$(find $O/cs -name "*.db" | head -1)
fi
[ -n "$DB" ] && python3 profiles/summarize_census.py $DB auto 6.5 2.3 > $O/census_isolated.md
rm -rf $O/cs
timeout 900 rocprofv3 --kernel-trace --pmc $SQ GRBM_GUI_ACTIVE -d $O/cp -- python3 bench.py --steps 8 --warmup 2 $LIGHT > $O/bench_pmc_pipelined.json 2>$O/cp.err
DB=$(find $O/cp -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python3 profiles/summarize_census.py $DB auto 6.5 2.3 > $O/census_pipelined.md
  python3 profiles/occupancy_timeline.py $DB > $O/census_pipelined_queue_occupancy.txt
fi
rm -rf $O/cp
head -30 $O/census_isolated.md | cut -c1-250
tail -5 $O/census_isolated.md
tail -5 $O/census_pipelined.md
head -8 $O/census_pipelined_queue_occupancy.txt
