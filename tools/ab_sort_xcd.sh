# A/B of the scatter kernel's workgroup -> XCD mapping (ZKR_SORT_XCD=0|1): isolated kernel time (ZKR_SERIAL=1, rocprofv3
# --kernel-trace --stats), WRITE_SIZE / FETCH_SIZE traffic in separate --pmc passes, then the pipelined proof rate of both.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/${1:-absort}; mkdir -p $O
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-js-baseline --no-pipeline --no-tx-circuit --no-bcast-modes --shards 0"
for v in 0 1; do
  export ZKR_SORT_XCD=$v
  ZKR_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/st$v -- python3 bench.py $ARGS > /dev/null 2>$O/st$v.err
  python3 profiles/summarize_rocpd.py $(find $O/st$v -name "*.db" | head -1) 0 > $O/serial_kernel_stats_xcd$v.md; rm -rf $O/st$v
  ZKR_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf$v -- python3 bench.py $ARGS > /dev/null 2>$O/pf$v.err
  ZKR_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw$v -- python3 bench.py $ARGS > /dev/null 2>$O/pw$v.err
  python3 profiles/summarize_pmc.py $(find $O/pf$v -name "*.db" | head -1) $(find $O/pw$v -name "*.db" | head -1) $O/pmc_traffic_xcd$v.json 20 3 > $O/pmc_traffic_xcd$v.md; rm -rf $O/pf$v $O/pw$v
  echo "== ZKR_SORT_XCD=$v"; grep -E "msm_scatter|msm_hist|msm_digits" $O/serial_kernel_stats_xcd$v.md; grep -E "msm_scatter|msm_hist|msm_digits" $O/pmc_traffic_xcd$v.md
done
for r in 1 2; do for v in 0 1; do
  ZKR_SORT_XCD=$v python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('xcd_map=$v round $r: %.2f proofs/s' % d['value'], {k: round(x,2) for k,x in d['stage_ms_per_proof'].items()})"
done; done
