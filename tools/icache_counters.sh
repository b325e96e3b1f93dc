cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/icache; mkdir -p $O
LIGHT="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0"
rocprofv3 --list-avail 2>/dev/null | grep -B3 "SQ_IFETCH_LEVEL, HIGH_RES" | head -8
for c in SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH; do
  ZKR_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/$c -- python3 bench.py --steps 2 --warmup 1 --no-pipeline $LIGHT > /dev/null 2>$O/$c.err
  DB=$(find $O/$c -name "*.db" | head -1)
  if [ -n "$DB" ]; then python3 profiles/summarize_counter.py $DB $c > $O/$c.md; else tail -3 $O/$c.err; fi
  rm -rf $O/$c
  head -16 $O/$c.md | cut -c1-170
done
