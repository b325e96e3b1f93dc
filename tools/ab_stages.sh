# per-stage times of two library builds on one box: bash tools/ab_stages.sh <other.so> <bench args...>
OTHER=$(realpath $1); shift
for v in shipped other shipped other; do
  if [ $v = other ]; then export ZKR_HIP_LIB=$OTHER; else unset ZKR_HIP_LIB; fi
  python bench.py --no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0 --no-tx-circuit "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$v %.2f proofs/s sclk %s W %s' % (d['value'], d['device_state_during_timed_region']['sclk_mhz_mean'], d['device_state_during_timed_region']['power_w_mean']), {k: round(x,2) for k,x in d['stage_ms_per_proof'].items()})"
done
