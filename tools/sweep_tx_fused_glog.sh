cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0 --steps 64"
for r in 1 2; do for v in "ZKR_MSM_GLOG=2" "ZKR_MSM_GLOG=3" "ZKR_MSM_GLOG=4" "ZKR_MSM_GLOG=5"; do
  env $v python3 bench.py $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] round $r:', round(d['tx_circuit']['proofs_per_s'],1), round(d['facade_pipeline_1024']['end_to_end_proofs_per_s'],1), round(d['facade_pipeline']['end_to_end_proofs_per_s'],1))"
done; done
