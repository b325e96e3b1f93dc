"""One proof at a time of the reference's tx circuit (BatchProcessTx(2, 6), m = 2^17): latency per synchronous call.
python tools/tx_single.py [n]   (under rocprofv3 --kernel-trace for the timeline of one proof: profiles/timeline.py)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "simple-zk-rollups_amd", "python"))
import torch
import zkr_hip
from zkr_hip import rollup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
circ = rollup.RollupCircuit(2, 6)
key, vk_bin = zkr_hip.ProvingKey.setup_r1cs(circ.r1cs(), device=0)
privs = [0x5A4B1000 + 7919 * i for i in range(8)]
state = rollup.RollupState(circ.depth)
for i, pv in enumerate(privs):
    state.deposit(i, rollup.gen_public_key(pv), 10 ** 20, 0)
txs = [state.transfer(j % 8, (j + 3) % 8, 10 ** 17 * (j + 1), 10 ** 15, privs[j % 8]) for j in range(circ.batch)]
wb = circ.calculate_witness(circ.flatten_inputs(state.batch_inputs(txs)))
w = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda(0)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    key.prove_device(w.data_ptr(), r=5, s=7, stream=stream)
ts = []
for i in range(n):
    t = time.perf_counter()
    key.prove_device(w.data_ptr(), r=11 + i, s=13 + i, stream=stream)
    ts.append(1e3 * (time.perf_counter() - t))
ts.sort()
print("single tx proof, device witness: median %.3f ms, min %.3f, max %.3f over %d calls" % (ts[len(ts) // 2], ts[0], ts[-1], n))
hw = bytes(wb)
ts = []
for i in range(n):
    t = time.perf_counter()
    key.prove(hw, 11 + i, 13 + i)
    ts.append(1e3 * (time.perf_counter() - t))
ts.sort()
print("single tx proof, host witness:   median %.3f ms, min %.3f, max %.3f" % (ts[len(ts) // 2], ts[0], ts[-1]))
