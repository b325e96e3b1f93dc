#!/bin/bash
# round 6, call 6: what slowed the ISOLATED accumulation by 5-7 % between round 5 and call 5's tree -- the pinned product order of
# add_mixed29 or the software-pipelined loop?  Isolated kernel durations (ZKR_SERIAL=1, one launch per table) of four builds + round 5.
# Then the FETCH_SIZE calibration on the gather pattern.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_06; mkdir -p $O
LIGHT="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --no-2-22 --no-withdraw --shards 0"
for v in r5 cur nopin oldloop nopin_oldloop; do
  ( [ $v != cur ] && export ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_$v.so; export ZKR_EXP_MERGE=0 ZKR_SERIAL=1
    rocprofv3 --kernel-trace --stats -d $O/s_$v -- python3 bench.py --steps 4 --warmup 1 --no-pipeline $LIGHT > /dev/null 2>$O/s_$v.err )
  python3 profiles/summarize_rocpd.py $(find $O/s_$v -name "*.db" | head -1) 0 > $O/serial_$v.md; rm -rf $O/s_$v
  echo "== $v"; grep -E 'msm_accum_kernel|msm_reduce1' $O/serial_$v.md | awk -F'|' '{printf "%-60s calls %s avg us %s\n", substr($2,1,60), $3, $5}'
done | tee $O/isolated_accum.txt
for cfg in "64 832 0 80" "128 832 0 80" "64 832 1 80"; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/g -- tools/bin/gather_bw $cfg > $O/g.out 2>$O/g.err
  DB=$(find $O/g -name "*.db" | head -1)
  echo "gather_bw $cfg: $(grep -v '^$' $O/g.out | tail -1)"; python3 profiles/summarize_counter.py $DB FETCH_SIZE | tail -2
  rm -rf $O/g
done | tee $O/fetch_size_calibration.txt
