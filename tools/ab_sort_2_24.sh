cd "${GRAFT_REPO_ROOT:?}" || exit 1
LIGHT="--no-cpu-baseline --no-tx-circuit --no-bcast-modes --shards 0 --no-js-baseline"
for v in "" "ZKR_SORT_NBL=8192 ZKR_MSM_J=16" "ZKR_SORT_NBL=8192 ZKR_MSM_J=64" "ZKR_MSM_J=16"; do
  echo "== [$v]"
  env $v python3 bench.py --log-m 24 --steps 6 --warmup 1 $LIGHT 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), {k: round(x,2) for k,x in d['stage_ms_per_proof'].items() if k in ('ntt','msm_sort','msm_accum_g1','msm_accum_g2')})"
done
