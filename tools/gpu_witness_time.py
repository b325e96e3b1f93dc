"""zkr_rollup_witness_batch_device: time per call for n batches of tx.circom (2 transactions each), and the pipeline
witness (GPU) -> prove_batch_device -> verify_batch.   python tools/gpu_witness_time.py [n_batches]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))
import torch, zkr_hip
from zkr_hip import rollup
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
circ = rollup.RollupCircuit(2, 6)
privs = [0x5A4B1000 + 7919 * i for i in range(8)]
state = rollup.RollupState(circ.depth)
for i, pv in enumerate(privs):
    state.deposit(i, rollup.gen_public_key(pv), 10 ** 24, 0)
flats = []
for b in range(n):
    txs = [state.transfer((2 * b + j) % 8, (2 * b + j + 3) % 8, 10 ** 15 * (j + 1) + b, 10 ** 12, privs[(2 * b + j) % 8]) for j in range(circ.batch)]
    flats.append(circ.flatten_inputs(state.batch_inputs(txs)))
circ.calculate_witness_batch_device(flats[:2])
for m in (1, 16, 64, n):
    t = time.perf_counter(); w = circ.calculate_witness_batch_device(flats[:m]); torch.cuda.synchronize(); el = time.perf_counter() - t
    print("GPU witness builder: %4d batches in %.1f ms (%.0f witnesses/s)" % (m, 1e3 * el, m / el))
assert bytes(w[n - 1].cpu().numpy().tobytes()) == circ.calculate_witness(flats[n - 1])
