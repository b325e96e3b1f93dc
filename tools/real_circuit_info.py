"""Key geometry, witness statistics and per-stage times of the real rollup circuit at the headline size (BatchProcessTx(18, 6)):
python3 tools/real_circuit_info.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "simple-zk-rollups_amd", "python"))
import torch, zkr_hip
from zkr_hip import rollup
c = rollup.RollupCircuit(18, 6)
key, vk = zkr_hip.ProvingKey.setup_r1cs(c.r1cs())
print(key.info(), key.windows())
privs = [0x5A4B1000 + 7919 * i for i in range(8)]
st = rollup.RollupState(6)
for i, pv in enumerate(privs): st.deposit(i, rollup.gen_public_key(pv), 10 ** 20, 0)
txs = [st.transfer(j % 8, (j + 3) % 8, 10 ** 17, 10 ** 15, privs[j % 8]) for j in range(18)]
wb = c.calculate_witness(st.batch_inputs(txs))
w = [int.from_bytes(wb[i:i+32], "little") for i in range(0, len(wb), 32)]
print("zeros", sum(1 for x in w if x == 0), "ones", sum(1 for x in w if x == 1), "small<2^64", sum(1 for x in w if 1 < x < 2**64), "of", len(w))
d = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda()
key.prove_batch_device([d.data_ptr()] * 4)
key.prof_enable(True); key.prof_reset()
n = 20
t = time.perf_counter(); key.prove_batch_device([d.data_ptr()] * n); torch.cuda.synchronize(); el = time.perf_counter() - t
print("ms per proof", 1e3 * el / n, {k: round(v[0] / n, 3) for k, v in key.prof().items()})
