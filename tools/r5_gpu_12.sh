#!/bin/bash
# round 5, last tree: long soak at 2^20, key lifecycle, the drop-in identity cost again
O=gpurun_out/r5_12b; mkdir -p $O
python tests/soak.py 20 240 > $O/soak_2_20.txt 2>&1; echo "soak rc=$?" >> $O/soak_2_20.txt
python tools/key_lifecycle.py 16 3 > $O/key_lifecycle.txt 2>&1; echo "lifecycle rc=$?" >> $O/key_lifecycle.txt
bash tools/r5_gpu_11.sh > /dev/null 2>&1; cp gpurun_out/r5_11b/dropin_identity_cost.txt $O/
tail -4 $O/soak_2_20.txt; tail -3 $O/key_lifecycle.txt; cat $O/dropin_identity_cost.txt
