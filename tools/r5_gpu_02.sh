#!/bin/bash
# round 5, second GPU call: the whole GPU suite on the tree with the scratch-free key-load / reduce3 kernels; key load before / after on
# one box; the bound of the 72-byte-row experiment (a build whose G1 accumulation takes a table point's words as limbs: garbage sums)
O=gpurun_out/r5_02; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log
for rep in 1 2; do
  ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_r5_head1.so python tools/key_load_time.py 20 2>&1 | sed 's/^/[before] /' >> $O/key_load.txt
  python tools/key_load_time.py 20 2>&1 | sed 's/^/[after ] /' >> $O/key_load.txt
done
for rep in 1 2 3; do
  python tools/rate_only.py 20 40 default >> $O/ab_fake_unpack.txt 2>&1
  ZKR_HIP_LIB=$PWD/tools/bin/libzkr_hip_fakeunpack.so timeout 300 python tools/rate_only.py 20 40 fake-unpack >> $O/ab_fake_unpack.txt 2>&1
done
tail -4 $O/tests_gpu.log; cat $O/key_load.txt $O/ab_fake_unpack.txt
