# Knob sweep at 2^22 (BASELINE configs[2]), pipelined rate, same box
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--log-m 22 --steps 16 --warmup 3 --no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0"
for v in "ZKR_UNUSED=0" "ZKR_MSM_GLOG=4" "ZKR_MSM_BIG=416" "ZKR_MSM_BIG=1664" "ZKR_MSM_J=16" "ZKR_SORT_NBL=8192" "ZKR_NTT_PRIO=2" "ZKR_UNUSED=1"; do
  env $v python3 bench.py $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v]:', round(d['value'],2))"
done
