cd "${GRAFT_REPO_ROOT:?}" || exit 1
for f in "--no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0" "--no-js-baseline --no-bcast-modes --shards 0" "--no-cpu-baseline --no-bcast-modes --shards 0" "--no-cpu-baseline --no-js-baseline"; do
  python3 bench.py $f 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$f]', round(d['value'],1), round(d['tx_circuit']['proofs_per_s'],1), round(d['facade_pipeline']['end_to_end_proofs_per_s'],1), round(d['dropin']['dropin_steady_ms'],2), round(d['rollup_circuit_2_20']['proofs_per_s'],1))"
done
