# Cheap sweep of plan knobs on the pipelined 2^20 headline (two rounds each, same box)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0"
for r in 1 2; do for v in "ZKR_UNUSED=0" "ZKR_MSM_GLOG=4" "ZKR_MSM_GLOG=6" "ZKR_RED_STREAMS=2" "ZKR_RED_STREAMS=4" "ZKR_MSM_BIG=128" "ZKR_MSM_BIG=512" "ZKR_NTT_PRIO=2" "ZKR_SORT_XCD=0" "ZKR_DIGITS_SPT=1"; do
  env $v python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] round $r:', round(d['value'],2))"
done; done
