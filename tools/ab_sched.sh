# A/B of the hand-over schedule (ZKR_SCHED) on one box: synchronous 2^20 proof, pipelined rate, single tx proof.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0"
for r in 1 2; do for v in "ZKR_SCHED=0" "ZKR_SCHED=1" "ZKR_SCHED=3"; do
  echo "== [$v] round $r: sync 2^20 / pipelined / tx single"
  env $v python3 tools/sync_single.py 20 20 2>/dev/null | tail -1
  env $v python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2))"
  env $v python3 tools/tx_single.py 30 2>&1 | grep "device witness"
done; done
