# Second knob sweep on the fused tx-circuit rate (bench.py tx_circuit leg), same box, two rounds
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0 --steps 64"
for r in 1 2; do for v in "ZKR_UNUSED=0" "ZKR_MSM_GLOG_FUSED=5" "ZKR_MSM_GLOG_FUSED=3" "ZKR_MSM_BIG=64" "ZKR_MSM_BIG=1024" "ZKR_NTT_PRIO=0" "ZKR_NTT_PRIO=2" "ZKR_SCHED=1" "ZKR_NO_MERGE_CH=1" "ZKR_RED_STREAMS=3" "ZKR_NO_SHARE_AC=1"; do
  env $v python3 bench.py $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] round $r:', round(d['tx_circuit']['proofs_per_s'],1), round(d['facade_pipeline_1024']['end_to_end_proofs_per_s'],1))"
done; done
