# A/B of compile-time variants (libraries under tools/bin, built with -D<knob>) on the pipelined 2^20 headline
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --shards 0"
for r in 1 2; do for lib in "" $(ls tools/bin/libzkr_hip_*.so); do
  v="ZKR_UNUSED=0"; [ -n "$lib" ] && v="ZKR_HIP_LIB=$lib"
  env $v python3 bench.py --steps 40 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[${lib:-default}] round $r:', round(d['value'],2))"
done; done
