# Two against three reduction streams on the REAL circuit filled to 2^20 (BatchProcessTx(18, 6): bit-heavy witness) and the tx circuit
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0"
for r in 1 2; do for v in "ZKR_RED_STREAMS=2" "ZKR_RED_STREAMS=3"; do
  env $v python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] round $r: synthetic', round(d['value'],2), 'BatchProcessTx(18,6)', round(d['rollup_circuit_2_20']['proofs_per_s'],2), 'tx fused', round(d['tx_circuit']['proofs_per_s'],1), 'pipeline', round(d['facade_pipeline']['end_to_end_proofs_per_s'],1))"
done; done
