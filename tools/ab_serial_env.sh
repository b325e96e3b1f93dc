# isolated stage times (ZKR_SERIAL=1) under environment variants on one box:  bash tools/ab_serial_env.sh "name:ENV=val,ENV=val" ...
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for r in 1 2; do for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS; export ZKR_SERIAL=1
    python3 bench.py --steps 6 --warmup 2 --no-pipeline --no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_proof']
print('%-10s round $r: accum_g1 %.3f  accum_g2 %.3f  reduce %.3f  ntt %.3f  sort %.3f  total %.2f ms  sclk %s' % ('$name', s['msm_accum_g1'], s['msm_accum_g2'], s['msm_reduce'], s['ntt'], s['msm_sort'], d['ms_per_step'], d['device_state_during_timed_region'].get('sclk_mhz_mean')))" )
done; done
