# A/B of C's oversized-bucket kernel on the auxiliary stream (ZKR_C_BIG_FIRST=1) against C's turn on its chain's stream (=0):
# fused tx-circuit throughput and the facade pipeline (five active streams against four)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0"
for r in 1 2 3; do for v in "ZKR_C_BIG_FIRST=1" "ZKR_C_BIG_FIRST=0"; do
  env $v python3 bench.py --steps 50 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] round $r', round(d['value'],2), round(d['tx_circuit']['proofs_per_s'],1), round(d['facade_pipeline']['end_to_end_proofs_per_s'],1), round(d['facade_pipeline_1024']['end_to_end_proofs_per_s'],1), round(d['dropin']['dropin_steady_ms'],3))"
done; done
