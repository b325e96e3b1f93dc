#!/bin/bash
# round 5, fourth GPU call: A and B1 reduced in one chain (default) against a chain per table (ZKR_NO_JOINT_AB=1), same box:
# one tx proof, one synchronous 2^20 proof, the pipelined 2^20 rate; then the whole GPU suite on the new default
O=gpurun_out/r5_04; mkdir -p $O
for r in 1 2 3; do for v in "ZKR_UNUSED=0" "ZKR_NO_JOINT_AB=1"; do
  echo -n "[$v] round $r: " >> $O/joint_ab.txt
  env $v python3 tools/tx_single.py 40 2>&1 | grep "witness" | cut -c17-60 | tr '\n' ' ' >> $O/joint_ab.txt
  env $v python3 tools/rate_only.py 20 40 x 2>&1 | grep proofs/s | cut -c5-48 | tr '\n' ' ' >> $O/joint_ab.txt
  env $v python3 tools/sync_single.py 20 12 2>&1 | grep "^synchronous" | cut -c1-45 >> $O/joint_ab.txt
done; done
cat $O/joint_ab.txt
python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log
tail -4 $O/tests_gpu.log
