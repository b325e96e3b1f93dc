# Same-box comparison of variants of the library / its knobs (run through gpurun):
#   bash tools/ab_variants.sh <rounds> "name:ENV=val,ENV=val" ...       (an empty assignment list = the shipped defaults)
# e.g. bash tools/ab_variants.sh 3 "shipped:" "ntt256:ZKR_HIP_LIB=tools/bin/libzkr_hip_ntt256.so" "acc1:ZKR_ACC_W_G1=1"
# The boxes of the pool differ by +-3 % among themselves: only numbers from one call are comparable.
N=$1; shift
for r in $(seq 1 $N); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
      [ -n "$ZKR_HIP_LIB" ] && export ZKR_HIP_LIB=$(realpath $ZKR_HIP_LIB)
      python bench.py --no-cpu-baseline --no-js-baseline --no-bcast-modes ${ZKR_AB_ARGS:-} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); v=d['roofline']['valu']
tx=(d.get('tx_circuit') or {}).get('proofs_per_s') or 0
print('%-10s round $r: %.2f proofs/s  tx %.1f  mul %.1f  sclk %s' % ('$name', d['value'], tx, v['peak_fq_mul_per_s_G'], d['device_state_during_timed_region']['sclk_mhz_mean']))" )
  done
done
