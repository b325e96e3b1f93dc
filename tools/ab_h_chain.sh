# A/B: the proof's last reduction chain (H, with C's sums) on the auxiliary stream (ZKR_H_CHAIN_AUX=1) or in the G1 chains' rotation
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0"
for r in 1 2 3; do for v in "ZKR_H_CHAIN_AUX=0" "ZKR_H_CHAIN_AUX=1"; do
  echo "== [$v] round $r: tx single / sync 2^20 / pipelined 2^20, tx fused"
  env $v python3 tools/tx_single.py 40 2>&1 | grep "device witness"
  env $v python3 tools/sync_single.py 20 20 2>/dev/null | tail -1
  env $v python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['tx_circuit']['proofs_per_s'],1))"
done; done
