cd "${GRAFT_REPO_ROOT:?}" || exit 1
for r in 1 2 3; do for v in "ZKR_SCHED=0" "ZKR_SCHED=1"; do
  echo "== [$v] round $r"
  env $v python3 tools/tx_single.py 40 2>&1 | grep "witness"
done; done
for v in "ZKR_SCHED=0" "ZKR_SCHED=1"; do
  env $v python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d.get('tx_circuit') or {}; print('$v', {k: (round(x,2) if isinstance(x,float) else x) for k,x in t.items() if not isinstance(x,(dict,list))})"
done
