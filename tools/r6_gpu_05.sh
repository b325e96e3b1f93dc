#!/bin/bash
# round 6, call 5: the software-pipelined accumulation loop (entry index two additions ahead, point one; G2: next point's lines touched)
# with the chain kernels at their new register footprint (cur) and at round 5's (fat: same instructions, clobbered high registers)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_05; mkdir -p $O
python -m pytest tests/test_gpu_stages.py -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log; tail -3 $O/tests_gpu.log
LIGHT="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --no-2-22 --no-withdraw --shards 0"
run() { local name=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    python3 bench.py --steps 30 --warmup 5 $LIGHT 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['config']; st=d['stage_ms_per_proof']
print('%-10s %.2f proofs/s  %.3f ms  sync %.2f ms  sclk %.0f  power %.0f W | acc_g1 %.2f acc_g2 %.2f ntt %.2f sort %.2f big %.2f reduce %.2f total %.2f' % ('$name', d['value'], d['ms_per_step'], b.get('sync_latency_ms') or 0, d['device_state_during_timed_region']['sclk_mhz_mean'], d['device_state_during_timed_region'].get('power_w_mean') or 0, st['msm_accum_g1'], st['msm_accum_g2'], st['ntt'], st['msm_sort'], st['msm_big'], st['msm_reduce'], st['total']))" ) | tee -a $O/ab_loop.txt
}
for r in 1 2; do
  run r5 ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so
  run cur_m0 ZKR_EXP_MERGE=0
  run fat_m0 ZKR_EXP_MERGE=0 ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_fat.so
  run cur_m3 ZKR_EXP_MERGE=3
  run fat_m3 ZKR_EXP_MERGE=3 ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_fat.so
done
ZKR_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/strace -- python3 bench.py --steps 4 --warmup 1 --no-pipeline $LIGHT > $O/bench_serial.json 2>$O/strace.err
python3 profiles/summarize_rocpd.py $(find $O/strace -name "*.db" | head -1) 0 > $O/serial_kernel_stats.md; rm -rf $O/strace
head -16 $O/serial_kernel_stats.md | cut -c1-150
