import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))
import torch, zkr_hip
from zkr_hip import rollup
circ = rollup.RollupCircuit()
key, vk = zkr_hip.ProvingKey.setup_r1cs(circ.r1cs())
privs = [0x5A4B1000 + 7919 * i for i in range(4)]
st = rollup.RollupState(circ.depth)
for i, pv in enumerate(privs):
    st.deposit(i, rollup.gen_public_key(pv), 10 ** 20, 0)
txs = [st.transfer(j, (j + 1) % 4, 10 ** 17, 10 ** 15, privs[j]) for j in range(circ.batch)]
wb = circ.calculate_witness(st.batch_inputs(txs))
d = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda()
key.prove_batch_device([d.data_ptr()] * 4)
torch.cuda.synchronize()
n = 200
ts, tc = 0.0, 0.0
t0 = time.perf_counter()
pend = []
for i in range(n):
    if len(pend) == 2:
        a = time.perf_counter(); key.prove_collect(pend.pop(0)); tc += time.perf_counter() - a
    a = time.perf_counter(); pend.append(key.prove_submit(d.data_ptr())); ts += time.perf_counter() - a
while pend:
    a = time.perf_counter(); key.prove_collect(pend.pop(0)); tc += time.perf_counter() - a
el = time.perf_counter() - t0
print("per proof %.3f ms: submit %.3f ms, collect (wait + host assembly) %.3f ms" % (1e3 * el / n, 1e3 * ts / n, 1e3 * tc / n))
