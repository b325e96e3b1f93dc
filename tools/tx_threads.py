"""One key of the tx circuit, T host threads each proving synchronously (zkr_prove_device): the library keeps two proofs
in flight, the threads overlap their host-side work (launch enqueue, proof assembly).  python3 tools/tx_threads.py T N"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))

import torch  # noqa: E402
import zkr_hip  # noqa: E402
from zkr_hip import rollup  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
circ = rollup.RollupCircuit()
key, vk = zkr_hip.ProvingKey.setup_r1cs(circ.r1cs())
privs = [0x5A4B1000 + 7919 * i for i in range(4)]
st = rollup.RollupState(circ.depth)
for i, pv in enumerate(privs):
    st.deposit(i, rollup.gen_public_key(pv), 10 ** 20, 0)
txs = [st.transfer(j, (j + 1) % 4, 10 ** 17, 10 ** 15, privs[j]) for j in range(circ.batch)]
wb = circ.calculate_witness(st.batch_inputs(txs))
d = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
ok = []


def work(count):
    ps = [key.prove_device(d.data_ptr(), 3, 5) for _ in range(count)]
    ok.append(all(p == ps[0] for p in ps))


for count in (4, n):
    ths = [threading.Thread(target=work, args=(count,)) for _ in range(T)]
    t = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    el = time.perf_counter() - t
assert all(ok)
print("%d threads: %.2f ms per proof (%.1f proofs/s)" % (T, 1e3 * el / (T * n), T * n / el))
