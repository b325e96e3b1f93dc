# Third sweep on the round's last tree (two reduction streams): pipelined 2^20 headline, two rounds each, same box
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0"
for r in 1 2; do for v in "ZKR_UNUSED=0" "ZKR_NTT_THREADS=256" "ZKR_ACC_PRIO=1" "ZKR_NO_PRIO=1" "ZKR_MSM_GLOG_G2=4" "ZKR_MSM_GLOG_G2=6" "ZKR_MSM_C=19" "ZKR_NO_SHARE_AC=1" "ZKR_ACC_W_G2=1" "ZKR_MSM_J=32"; do
  env $v python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] round $r:', round(d['value'],2))"
done; done
