# Same-box comparison of variants of the library / its knobs (run through gpurun):
#   bash tools/ab_variants.sh <rounds> "name:ENV=val,ENV=val" ...       (an empty assignment list = the shipped defaults)
# e.g. bash tools/ab_variants.sh 3 "shipped:" "r5:ZKR_HIP_LIB=tools/bin/libzkr_hip_r5.so" "c19:ZKR_MSM_C=19"   (library builds under tools/bin/, or the few knobs the library keeps: INTEGRATION.md)
# The boxes of the pool differ by +-3 % among themselves: only numbers from one call are comparable.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
N=$1; shift
for r in $(seq 1 $N); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
      [ -n "$ZKR_HIP_LIB" ] && export ZKR_HIP_LIB=$(realpath $ZKR_HIP_LIB)
      python bench.py --no-cpu-baseline --no-js-baseline --no-bcast-modes --shards 0 ${ZKR_AB_ARGS:-} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); v=d['roofline']['valu']
tx=(d.get('tx_circuit') or {}).get('proofs_per_s') or 0
b=d['config'].get('boundary') or {}
st=d['stage_ms_per_proof']
print('%-10s round $r: %.2f proofs/s  sync %.2f ms  tx fused %.1f/s  tx single %s ms  dropin %s ms  pipe1024 %s/s  | sort %.2f ntt %.2f spmv %.2f | mul %.1f  sclk %s' % ('$name', d['value'], b.get('sync_latency_ms') or 0, tx,
      b.get('tx_circuit_single_proof_ms') and round(b['tx_circuit_single_proof_ms'], 3), b.get('tx_circuit_dropin_call_ms') and round(b['tx_circuit_dropin_call_ms'], 3),
      b.get('tx_circuit_facade_pipeline_proofs_per_s') and round(b['tx_circuit_facade_pipeline_proofs_per_s'], 1),
      st['msm_sort'], st['ntt'], st['spmv'], v['peak_fq_mul_per_s_G'], d['device_state_during_timed_region']['sclk_mhz_mean']))" )
  done
done
