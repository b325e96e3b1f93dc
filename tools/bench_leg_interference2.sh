# The bench's side legs after the replication leg has created and freed two keys: shared device streams (default) against a
# stream set per key (ZKR_PRIVATE_STREAMS=1), and the latter with eight hardware queues
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for v in "ZKR_UNUSED=0" "ZKR_PRIVATE_STREAMS=1" "ZKR_PRIVATE_STREAMS=1 GPU_MAX_HW_QUEUES=8"; do
  env $v python3 bench.py --no-cpu-baseline --no-js-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v]', round(d['value'],1), round(d['tx_circuit']['proofs_per_s'],1), round(d['facade_pipeline']['end_to_end_proofs_per_s'],1), round(d['dropin']['dropin_steady_ms'],2), round(d['rollup_circuit_2_20']['proofs_per_s'],1))"
done
