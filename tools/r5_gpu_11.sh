#!/bin/bash
# round 5: the key cache's compare-based identity on the GPU paths that use it (Python facade, Node host, drop-in legs)
O=gpurun_out/r5_11b; mkdir -p $O
python -m pytest tests/test_gpu_multi.py tests/test_node_host.py tests/test_gpu_rollup.py tests/test_gpu_stages.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
python - <<'P' > $O/dropin_identity_cost.txt 2>&1
import os, sys, time
sys.path.insert(0, "simple-zk-rollups_amd/python")
import zkr_hip
from zkr_hip import rollup
circ = rollup.RollupCircuit(2, 6)
pkb, vk = zkr_hip.setup_r1cs_websnark(circ.r1cs(), device=0)
privs = [0x5A4B1000 + 7919 * i for i in range(4)]
st = rollup.RollupState(circ.depth)
for i, pv in enumerate(privs):
    st.deposit(i, rollup.gen_public_key(pv), 10 ** 20, 0)
txs = [st.transfer(j % 4, (j + 1) % 4, 10 ** 17, 10 ** 15, privs[j % 4]) for j in range(circ.batch)]
wb = circ.calculate_witness(circ.flatten_inputs(st.batch_inputs(txs)))
for mode, fresh in (("same buffer object per call", False), ("NEW buffer object per call (common.ts:28)", True)):
    zkr_hip.clear_key_cache()
    ts = []
    for i in range(10):
        buf = bytes(bytearray(pkb)) if fresh else pkb
        t = time.perf_counter()
        zkr_hip.build_bn128(0).groth16GenProof(wb, buf)
        ts.append(1e3 * (time.perf_counter() - t))
    print("%s: first %.1f ms, then median %.2f ms (%d MB key); compares so far %d" % (mode, ts[0], sorted(ts[1:])[4], len(pkb) >> 20, zkr_hip.key_cache_stats["compares"]))
P
cat $O/dropin_identity_cost.txt
