# Sweep of the HIP runtime's hardware-queue count (GPU_MAX_HW_QUEUES, default 4): headline, fused tx circuit, facade pipeline
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0"
for r in 1 2; do for q in ${AB_QUEUES:-4 5 6 8}; do
  GPU_MAX_HW_QUEUES=$q python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('queues $q round $r:', round(d['value'],2), round(d['tx_circuit']['proofs_per_s'],1), round(d['facade_pipeline']['end_to_end_proofs_per_s'],1), round(d['facade_pipeline_1024']['end_to_end_proofs_per_s'],1), round(d['dropin']['dropin_steady_ms'],2))"
done; done
python3 tools/tx_single.py 40 2>&1 | grep "witness"
