# SQ counters per kernel at the benchmarked size (2^20), isolated kernels (ZKR_SERIAL=1): one --pmc pass per counter
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/${1:-r2pmc}; mkdir -p $O
for c in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU; do
  ZKR_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-tx-circuit --no-bcast-modes --shards 0 > /dev/null 2>$O/$c.err
  DB=$(find $O/$c -name "*.db" | head -1)
  if [ -n "$DB" ]; then python3 profiles/summarize_counter.py $DB $c > $O/$c.md; else tail -3 $O/$c.err; fi
  rm -rf $O/$c
done
ls -la $O; head -12 $O/SQ_ACTIVE_INST_VALU.md
