cd "${GRAFT_REPO_ROOT:?}" || exit 1
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
B="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0"
for r in 1 2; do
for v in "" "ZKR_HIP_LIB=tools/bin/libzkr_hip_prep3.so" "ZKR_HIP_LIB=tools/bin/libzkr_hip_prep3.so ZKR_NTT_PRIO=3" "ZKR_NTT_PRIO=3"; do
  echo "== [$v] pipelined / sync"
  env $v python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), {k: round(x,2) for k,x in d['stage_ms_per_proof'].items() if k in ('ntt','msm_sort','msm_accum_g1')})"
  env $v python3 bench.py --no-pipeline --steps 30 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"
done; done
for v in "" "ZKR_HIP_LIB=tools/bin/libzkr_hip_prep3.so ZKR_NTT_PRIO=3"; do echo "== [$v] tx"; env $v python tools/tx_single.py 30 2>&1 | grep -v amdgpu.ids | head -2; done
