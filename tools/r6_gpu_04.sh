#!/bin/bash
# round 6, call 4: the oversized-bucket / reduction kernels built to FIT beside the accumulations (G1 forms 136 VGPRs, G2 forms 248;
# before 160-198 / 314-394) -- parity of the stages they touch, then same-box A/B against round 5's library with the accumulations
# launched per table (merge0) and with B1 + A + C in one launch (merge3)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_04; mkdir -p $O
python -m pytest tests/test_gpu_stages.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log; tail -4 $O/tests_gpu.log
LIGHT="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --no-2-22 --no-withdraw --shards 0"
run() { local name=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    python3 bench.py --steps 30 --warmup 5 $LIGHT 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['config']; st=d['stage_ms_per_proof']
print('%-10s %.2f proofs/s  %.3f ms  sync %.2f ms  sclk %.0f  power %.0f W | acc_g1 %.2f acc_g2 %.2f ntt %.2f sort %.2f big %.2f reduce %.2f total %.2f' % ('$name', d['value'], d['ms_per_step'], b.get('sync_latency_ms') or 0, d['device_state_during_timed_region']['sclk_mhz_mean'], d['device_state_during_timed_region'].get('power_w_mean') or 0, st['msm_accum_g1'], st['msm_accum_g2'], st['ntt'], st['msm_sort'], st['msm_big'], st['msm_reduce'], st['total']))" ) | tee -a $O/ab_fit.txt
}
for r in 1 2 3; do
  run r5 ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so
  run fit_merge0 ZKR_EXP_MERGE=0
  run fit_merge3 ZKR_EXP_MERGE=3
done
run fit_merge1 ZKR_EXP_MERGE=1
run fit_merge2 ZKR_EXP_MERGE=2
# the reference's tx circuit (fused batches, single proof): the chain kernels are its critical path
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-js-baseline --no-bcast-modes --no-2-22 --no-withdraw --shards 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('new  tx single %.3f ms  tx fused %.1f/s  dropin %.3f ms' % (c['tx_single_proof_ms'], c['tx_fused_proofs_per_s'], c['tx_dropin_call_ms']))" | tee -a $O/ab_fit.txt
ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-js-baseline --no-bcast-modes --no-2-22 --no-withdraw --shards 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('r5   tx single %.3f ms  tx fused %.1f/s  dropin %.3f ms' % (c['tx_single_proof_ms'], c['tx_fused_proofs_per_s'], c['tx_dropin_call_ms']))" | tee -a $O/ab_fit.txt
