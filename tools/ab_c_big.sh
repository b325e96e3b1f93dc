# A/B: C's oversized-bucket kernel in front of every chain (ZKR_C_BIG_FIRST=1) or with C's turn (=0)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0"
for r in 1 2 3; do for v in "ZKR_C_BIG_FIRST=1" "ZKR_C_BIG_FIRST=0"; do
  echo "== [$v] round $r: tx single / sync 2^20 / pipelined"
  env $v python3 tools/tx_single.py 40 2>&1 | grep "witness"
  env $v python3 tools/sync_single.py 20 20 2>/dev/null | tail -1
  env $v python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2))"
done; done

