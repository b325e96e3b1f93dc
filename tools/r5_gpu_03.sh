#!/bin/bash
# round 5, third GPU call: CU-reserve experiment (the accumulation stream kept off the last n CUs), alone and with the early hand-over
# of the accumulations (ZKR_SCHED=1): latency of one tx proof, of one synchronous 2^20 proof, and the pipelined 2^20 rate; same box
O=gpurun_out/r5_03; mkdir -p $O
for r in 1 2; do for v in "ZKR_UNUSED=0" "ZKR_SCHED=1" "ZKR_ACC_CU_RESERVE=16" "ZKR_ACC_CU_RESERVE=32" "ZKR_ACC_CU_RESERVE=16 ZKR_SCHED=1" "ZKR_ACC_CU_RESERVE=32 ZKR_SCHED=1" "ZKR_ACC_CU_RESERVE=64 ZKR_SCHED=1"; do
  echo -n "[$v] round $r: " >> $O/cu_reserve_tx.txt; env $v python3 tools/tx_single.py 40 2>&1 | grep "witness" | cut -c17-90 | tr '\n' ' ' >> $O/cu_reserve_tx.txt; echo >> $O/cu_reserve_tx.txt
done; done
for r in 1 2; do for v in "ZKR_UNUSED=0" "ZKR_ACC_CU_RESERVE=16" "ZKR_ACC_CU_RESERVE=32" "ZKR_ACC_CU_RESERVE=32 ZKR_SCHED=1"; do
  echo -n "[$v] round $r: " >> $O/cu_reserve_2_20.txt
  env $v python3 tools/rate_only.py 20 40 x 2>&1 | grep proofs/s | cut -c5-120 | tr '\n' ' ' >> $O/cu_reserve_2_20.txt
  env $v python3 tools/sync_single.py 20 12 2>&1 | grep "synchronous" | cut -c1-60 >> $O/cu_reserve_2_20.txt
done; done
cat $O/cu_reserve_tx.txt $O/cu_reserve_2_20.txt
