#!/bin/bash
# round 6, call 2: the three w-driven G1 accumulations in ONE launch -- GPU suite on the new tree, then same-box
# A/B against round 5's library (tools/bin/libzkr_hip_r5.so), three alternating rounds; the old library's concurrent-accumulation
# schedule (ZKR_SCHED=2) beside them for the record
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_02; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log; tail -4 $O/tests_gpu.log
LIGHT="--no-cpu-baseline --no-js-baseline --no-tx-circuit --no-bcast-modes --no-2-22 --no-withdraw --shards 0"
run() { # name, env...
  local name=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    python3 bench.py --steps 40 --warmup 5 $LIGHT 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['config']
print('%-14s %.2f proofs/s  %.3f ms  sync %.2f ms  hostbuf %.1f/s  sclk %s  acc_g1 launch %.1f us' % ('$name', d['value'], d['ms_per_step'], b.get('sync_latency_ms') or 0, b.get('host_buffer_sync_proofs_per_s') or 0, d['device_state_during_timed_region']['sclk_mhz_mean'], 1e3 * d['roofline']['avg_launch_ms']))" ) | tee -a $O/ab_merged_launch.txt
}
for r in 1 2 3; do
  run r5 ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so
  run merged
done
run r5_sched2 ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so ZKR_SCHED=2
run r5_sched3 ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5.so ZKR_SCHED=3
python3 - > $O/key_info_2_20.txt <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "simple-zk-rollups_amd", "python"))
import zkr_hip
key, wb, aux = zkr_hip.ProvingKey.synth(20, 73, 0x5A4B0001, 0x5A4B00FF)
print(key.info())
PY
cat $O/key_info_2_20.txt
