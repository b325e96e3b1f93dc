cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/pmc22; mkdir -p $O
for c in SQ_INSTS_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU; do
  ZKR_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/$c -- python3 bench.py --log-m 22 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-tx-circuit > /dev/null 2>$O/$c.err
  DB=$(find $O/$c -name "*.db" | head -1)
  if [ -n "$DB" ]; then python3 profiles/summarize_counter.py $DB $c > $O/$c.md; else tail -3 $O/$c.err; fi
  rm -rf $O/$c
done
ls -la $O; head -8 $O/SQ_INSTS_VALU.md
