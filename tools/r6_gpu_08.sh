#!/bin/bash
# round 6, call 8: point feeds of the accumulation -- G2 through the LDS DMA path, G1 one / two points ahead in registers; the chain
# kernels in their simple form (pinned product order only).  Parity first, then isolated launch times and pipelined rates.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_08; mkdir -p $O
python -m pytest tests/test_gpu_stages.py -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log; tail -3 $O/tests_gpu.log
for r in 1 2; do
  ( export ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so; ZKR_SERIAL=1 python3 tools/rate_only.py 20 12 serial_r5; python3 tools/rate_only.py 20 40 pipelined_r5 )
  for f in 1 2; do for m in 0 3; do
    ( export ZKR_EXP_FEED=$f ZKR_EXP_MERGE=$m; [ $m = 0 ] && ZKR_SERIAL=1 python3 tools/rate_only.py 20 12 serial_feed${f}; python3 tools/rate_only.py 20 40 pipelined_feed${f}_merge${m} )
  done; done
done 2>&1 | grep '^\[' | tee $O/feeds.txt
LIGHT="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --no-2-22 --no-withdraw --shards 0"
run() { local name=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    python3 bench.py --steps 30 --warmup 5 $LIGHT 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['config']; st=d['stage_ms_per_proof']
print('%-12s %.2f proofs/s  %.3f ms  sync %.2f ms  sclk %.0f  power %.0f W | acc_g1 %.2f acc_g2 %.2f ntt %.2f sort %.2f big %.2f reduce %.2f total %.2f' % ('$name', d['value'], d['ms_per_step'], b.get('sync_latency_ms') or 0, d['device_state_during_timed_region']['sclk_mhz_mean'], d['device_state_during_timed_region'].get('power_w_mean') or 0, st['msm_accum_g1'], st['msm_accum_g2'], st['ntt'], st['msm_sort'], st['msm_big'], st['msm_reduce'], st['total']))" ) | tee -a $O/ab_feeds_bench.txt
}
for r in 1 2; do
  run r5 ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so
  run feed1_m0 ZKR_EXP_FEED=1 ZKR_EXP_MERGE=0
  run feed2_m0 ZKR_EXP_FEED=2 ZKR_EXP_MERGE=0
  run feed2_m3 ZKR_EXP_FEED=2 ZKR_EXP_MERGE=3
done
