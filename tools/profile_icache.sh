cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/${1:-r2ic}; mkdir -p $O
for c in SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY; do
  ZKR_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-tx-circuit --no-bcast-modes --shards 0 > /dev/null 2>$O/$c.err
  DB=$(find $O/$c -name "*.db" | head -1)
  if [ -n "$DB" ]; then python3 profiles/summarize_counter.py $DB $c > $O/$c.md; else tail -3 $O/$c.err; fi
  rm -rf $O/$c
done
