# A/B of environment settings on the sort kernels:  bash tools/ab_env.sh <tag> "<VAR=.. VAR=..>" "<VAR=..>" ...
# For every setting: isolated kernel times (ZKR_SERIAL=1, rocprofv3 --kernel-trace --stats), WRITE_SIZE / FETCH_SIZE traffic in
# separate --pmc passes; then the pipelined proof rate of every setting, two rounds.  "-" = the defaults.  AB_BENCH_ARGS="--log-m 22"
# AB_STEPS=20: another size.  AB_NO_RATE=1: kernel times and traffic only.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/${1:-abenv}; mkdir -p $O; shift
ARGS="$AB_BENCH_ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-js-baseline --no-pipeline --no-tx-circuit --no-bcast-modes --shards 0"
i=0
for v in "$@"; do
  [ "$v" = "-" ] && v=""
  ( [ -n "$v" ] && export $v; export ZKR_SERIAL=1
    rocprofv3 --kernel-trace --stats -d $O/st$i -- python3 bench.py $ARGS > /dev/null 2>$O/st$i.err
    python3 profiles/summarize_rocpd.py $(find $O/st$i -name "*.db" | head -1) 0 > $O/serial_kernel_stats_$i.md; rm -rf $O/st$i
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf$i -- python3 bench.py $ARGS > /dev/null 2>$O/pf$i.err
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw$i -- python3 bench.py $ARGS > /dev/null 2>$O/pw$i.err
    python3 profiles/summarize_pmc.py $(find $O/pf$i -name "*.db" | head -1) $(find $O/pw$i -name "*.db" | head -1) $O/pmc_traffic_$i.json 30 3 > $O/pmc_traffic_$i.md; rm -rf $O/pf$i $O/pw$i )
  echo "== [$i] $v"; grep -E "msm_scatter|msm_hist|msm_digits|colscan" $O/serial_kernel_stats_$i.md | cut -c1-150; grep -E "msm_scatter|msm_hist|msm_digits|colscan" $O/pmc_traffic_$i.md
  i=$((i+1))
done
[ -n "$AB_NO_RATE" ] && exit 0
for r in 1 2; do for v in "$@"; do
  [ "$v" = "-" ] && v=""
  ( [ -n "$v" ] && export $v; python3 bench.py $AB_BENCH_ARGS --steps ${AB_STEPS:-40} --warmup 5 --no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] round $r: %.2f proofs/s' % d['value'], {k: round(x,2) for k,x in d['stage_ms_per_proof'].items()})" )
done; done
