#!/bin/bash
# round 5: early hand-over with the G2 accumulation built for ONE wavefront per SIMD (registers left for the preparation chain's kernels)
O=gpurun_out/r5_07; mkdir -p $O
for r in 1 2 3; do for v in "ZKR_UNUSED=0" "ZKR_SCHED=1" "ZKR_SCHED=1 ZKR_ACC_SPLIT_W_G2=1" "ZKR_ACC_SPLIT_W_G2=1" "ZKR_SCHED=1 ZKR_ACC_SPLIT_W_G2=1 ZKR_ACC_SPLIT=4"; do
  echo -n "[$v] round $r: " >> $O/sched_g2w1.txt
  env $v python3 tools/tx_single.py 40 2>&1 | grep "witness" | cut -c17-60 | tr '\n' ' ' >> $O/sched_g2w1.txt; echo >> $O/sched_g2w1.txt
done; done
cat $O/sched_g2w1.txt
