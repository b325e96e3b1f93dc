# stage sums of the default bench at the given args: bash tools/stages.sh <bench args...>
python bench.py --no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0 --no-tx-circuit "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('%.2f proofs/s %.2f ms/proof sclk %s W %s' % (d['value'], d['ms_per_step'], d['device_state_during_timed_region']['sclk_mhz_mean'], d['device_state_during_timed_region']['power_w_mean']), {k: round(x,2) for k,x in d['stage_ms_per_proof'].items()})"
