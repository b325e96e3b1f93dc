"""Throughput of the reference's tx circuit (BatchProcessTx(2, 6), m = 2^17) against the batch size handed to one
zkr_prove_batch_device call, with the per-stage GPU times of the library's own profiler.  python tools/tx_profile.py"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "simple-zk-rollups_amd", "python"))
import torch
import zkr_hip
from zkr_hip import rollup

circ = rollup.RollupCircuit(2, 6)
key, vk_bin = zkr_hip.ProvingKey.setup_r1cs(circ.r1cs(), device=0)
privs = [0x5A4B1000 + 7919 * i for i in range(8)]
state = rollup.RollupState(circ.depth)
for i, pv in enumerate(privs):
    state.deposit(i, rollup.gen_public_key(pv), 10 ** 20, 0)
wits, pubs = [], []
for b in range(4):
    txs = [state.transfer((2 * b + j) % 8, (2 * b + j + 3) % 8, 10 ** 17 * (j + 1), 10 ** 15, privs[(2 * b + j) % 8]) for j in range(circ.batch)]
    wb = circ.calculate_witness(circ.flatten_inputs(state.batch_inputs(txs)))
    pubs.append(circ.public_signals(wb))
    wits.append(torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda(0))
stream = torch.cuda.current_stream().cuda_stream
print("fuse", key.fuse(), "info", key.info())
out = {}
for n in [int(x) for x in os.environ.get("TX_COUNTS", "1,2,8,16,50,64,128,256").split(",")]:
    ptrs = [wits[i % 4].data_ptr() for i in range(n)]
    key.prove_batch_device(ptrs, stream=stream)
    torch.cuda.synchronize()
    key.prof_enable(True); key.prof_reset()
    t = time.perf_counter()
    proofs = key.prove_batch_device(ptrs, stream=stream)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    prof = key.prof(); key.prof_enable(False)
    assert zkr_hip.verify_batch(vk_bin, proofs, [pubs[i % 4] for i in range(n)])
    out[n] = {"proofs_per_s": n / el, "ms_per_proof": 1e3 * el / n, "stage_ms_per_proof": {k: v[0] / n for k, v in prof.items()} if isinstance(prof, dict) else prof}
    print(n, "proofs: %.1f proofs/s  %.3f ms/proof" % (n / el, 1e3 * el / n), {k: round(v, 3) for k, v in out[n]["stage_ms_per_proof"].items()} if isinstance(out[n]["stage_ms_per_proof"], dict) else "")
