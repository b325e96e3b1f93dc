cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0"
for r in 1 2 3; do for v in "ZKR_RED_STREAMS=3" "ZKR_RED_STREAMS=2" "ZKR_RED_STREAMS=1" "ZKR_RED_STREAMS=2 ZKR_MSM_BIG=128" "ZKR_RED_STREAMS=2 ZKR_MSM_BIG=64" "ZKR_RED_STREAMS=2 ZKR_NTT_PRIO=2"; do
  env $v python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] round $r:', round(d['value'],2))"
done; done
for v in "ZKR_RED_STREAMS=3" "ZKR_RED_STREAMS=2"; do
  env $v python3 tools/sync_single.py 20 20 2>/dev/null | tail -1
  env $v python3 bench.py --log-m 22 --steps 16 --warmup 3 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] 2^22:', round(d['value'],2))"
done
