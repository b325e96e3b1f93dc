# A/B of the HIP runtime's hardware-queue count (GPU_MAX_HW_QUEUES, default 4; a key keeps 6-7 streams busy)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0"
for r in 1 2; do for v in "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8 ZKR_H_CHAIN_AUX=1" "GPU_MAX_HW_QUEUES=8 ZKR_RED_STREAMS=3"; do
  echo "== [$v] round $r: tx single / sync 2^20 / pipelined 2^20, tx fused"
  env $v python3 tools/tx_single.py 40 2>&1 | grep "device witness"
  env $v python3 tools/sync_single.py 20 20 2>/dev/null | tail -1
  env $v python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['tx_circuit']['proofs_per_s'],1))"
done; done
