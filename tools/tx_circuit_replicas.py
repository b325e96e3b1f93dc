"""Small circuits leave the GPU under-filled (latency-bound chains): R key replicas of the tx circuit on one GPU, one host
thread each, every replica with two proofs in flight.  python3 tools/tx_circuit_replicas.py <replicas> <proofs per replica>"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))

import torch  # noqa: E402
import zkr_hip  # noqa: E402
from zkr_hip import rollup  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
circ = rollup.RollupCircuit()
tox = [11, 22, 33, 44, 55]
keys = [zkr_hip.ProvingKey.setup_r1cs(circ.r1cs(), toxic=tox)[0] for _ in range(R)]
privs = [0x5A4B1000 + 7919 * i for i in range(4)]
st = rollup.RollupState(circ.depth)
for i, pv in enumerate(privs):
    st.deposit(i, rollup.gen_public_key(pv), 10 ** 20, 0)
txs = [st.transfer(j, (j + 1) % 4, 10 ** 17, 10 ** 15, privs[j]) for j in range(circ.batch)]
wb = circ.calculate_witness(st.batch_inputs(txs))
d = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
out = [None] * R


def work(j, count):
    out[j] = keys[j].prove_batch_device([d.data_ptr()] * count, rs=[7] * count, ss=[9] * count)


for count in (4, n):
    ths = [threading.Thread(target=work, args=(j, count)) for j in range(R)]
    t = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    el = time.perf_counter() - t
assert all(o[-1] == out[0][0] for o in out)
print("%d replicas: %.2f ms per proof (%.1f proofs/s)" % (R, 1e3 * el / (R * n), R * n / el))
