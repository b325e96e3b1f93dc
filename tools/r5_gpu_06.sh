#!/bin/bash
# round 5: with A and B1 in one chain, is handing the accumulations over early (ZKR_SCHED=1) still a loss for ONE tx proof?  same box
O=gpurun_out/r5_06; mkdir -p $O
for r in 1 2 3; do for v in "ZKR_UNUSED=0" "ZKR_SCHED=1"; do
  echo -n "[$v] round $r: " >> $O/sched_joint.txt
  env $v python3 tools/tx_single.py 40 2>&1 | grep "witness" | cut -c17-60 | tr '\n' ' ' >> $O/sched_joint.txt
  env $v python3 tools/sync_single.py 20 12 2>&1 | grep "^synchronous" | cut -c1-45 >> $O/sched_joint.txt
done; done
python -m pytest tests/test_gpu_shard.py -m gpu -x -q 2>&1 | tail -2 >> $O/sched_joint.txt
cat $O/sched_joint.txt
