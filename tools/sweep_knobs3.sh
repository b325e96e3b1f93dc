# Second sweep on top of two reduction streams (pipelined 2^20 headline, two rounds each, same box)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0"
for r in 1 2; do for v in "ZKR_UNUSED=0" "ZKR_MSM_GLOG=4" "ZKR_MSM_GLOG=6" "ZKR_SCHED=1" "ZKR_H_CHAIN_AUX=1" "ZKR_NO_MERGE_CH=1" "ZKR_ACC_W_G1=3" "ZKR_NTT_PRIO=0" "ZKR_NTT_PRIO=3" "ZKR_C_BIG_FIRST=1"; do
  env $v python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] round $r:', round(d['value'],2))"
done; done
