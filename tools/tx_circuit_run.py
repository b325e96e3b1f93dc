"""Proves the reference's tx circuit (BatchProcessTx(2, 6), m = 2^17) N times on cuda:0: a small driver to put under
rocprofv3 (`rocprofv3 --kernel-trace --stats -d out -- python3 tools/tx_circuit_run.py 50`).  ZKR_SERIAL=1 for isolated
kernel times."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))

import torch  # noqa: E402
import zkr_hip  # noqa: E402
from zkr_hip import rollup  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
circ = rollup.RollupCircuit()
key, vk = zkr_hip.ProvingKey.setup_r1cs(circ.r1cs())
privs = [0x5A4B1000 + 7919 * i for i in range(4)]
st = rollup.RollupState(circ.depth)
for i, pv in enumerate(privs):
    st.deposit(i, rollup.gen_public_key(pv), 10 ** 20, 0)
txs = [st.transfer(j, (j + 1) % 4, 10 ** 17, 10 ** 15, privs[j]) for j in range(circ.batch)]
wb = circ.calculate_witness(st.batch_inputs(txs))
d = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda()
stream = torch.cuda.current_stream().cuda_stream
key.prove_batch_device([d.data_ptr()] * 4, stream=stream)
torch.cuda.synchronize()
for mode, depth in (("pipelined", None), ("synchronous", 1)):
    t = time.perf_counter()
    proofs = key.prove_batch_device([d.data_ptr()] * n, stream=stream, depth=depth)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    print("%s: %.2f ms per proof (%.1f proofs/s)" % (mode, 1e3 * el / n, n / el))
assert zkr_hip.verify(vk, proofs[0], circ.public_signals(wb))
print("windows", key.windows(), "info", key.info())
