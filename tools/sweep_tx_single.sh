# Knob sweep on the latency of ONE tx-circuit proof (tools/tx_single.py, device witness), same box
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for r in 1 2; do for v in "ZKR_UNUSED=0" "ZKR_MSM_GLOG=1" "ZKR_MSM_GLOG=3" "ZKR_NTT_THREADS=256" "ZKR_MSM_J=8" "ZKR_MSM_J=32" "ZKR_SORT_NBL=1024" "ZKR_SORT_NBL=8192" "ZKR_DIGITS_SPT=1" "ZKR_DIGITS_SPT=4" "ZKR_MSM_BIG=128" "ZKR_MSM_BIG=1024" "ZKR_C_BIG_FIRST=0"; do
  echo -n "[$v] round $r: "; env $v python3 tools/tx_single.py 40 2>&1 | grep "device witness" | cut -c33-80
done; done
