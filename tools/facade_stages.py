"""Stage rates of the facade pipeline one by one (tx circuit): witness pool alone, zkr_prove_batch alone, zkr_verify_batch alone."""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))
from concurrent.futures import ThreadPoolExecutor
import bench, zkr_hip
from zkr_hip import rollup
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
circ = rollup.RollupCircuit(2, 6)
key, vk_bin = zkr_hip.ProvingKey.setup_r1cs(circ.r1cs(), device=0)
privs = [0x5A4B1000 + 7919 * i for i in range(8)]
state = rollup.RollupState(circ.depth)
for i, pv in enumerate(privs):
    state.deposit(i, rollup.gen_public_key(pv), 10 ** 24, 0)
flats = []
for b in range(n):
    txs = [state.transfer((2 * b + j) % 8, (2 * b + j + 3) % 8, 10 ** 15 * (j + 1) + b, 10 ** 12, privs[(2 * b + j) % 8]) for j in range(circ.batch)]
    flats.append(circ.flatten_inputs(state.batch_inputs(txs)))
t = time.perf_counter(); w1 = circ.calculate_witness(flats[0]); print("one witness call: %.2f ms" % (1e3 * (time.perf_counter() - t)))
for workers in (4, 8, 14, 16, 32):
    t = time.perf_counter()
    with ThreadPoolExecutor(workers) as pool:
        wits = list(pool.map(circ.calculate_witness, flats))
    el = time.perf_counter() - t
    print("witness pool %2d threads: %.0f witnesses/s" % (workers, n / el))
key.prove_batch(wits[:8])
t = time.perf_counter(); proofs = key.prove_batch(wits); el = time.perf_counter() - t
print("zkr_prove_batch of %d host witnesses: %.0f proofs/s" % (n, n / el))
t = time.perf_counter(); pubs = [circ.public_signals(w) for w in wits]; print("public_signals: %.2f ms each" % (1e3 * (time.perf_counter() - t) / n))
t = time.perf_counter(); ok = zkr_hip.verify_batch(vk_bin, proofs, pubs); el = time.perf_counter() - t
print("zkr_verify_batch: %.0f proofs/s (%s)" % (n / el, ok))
t = time.perf_counter(); ok = zkr_hip.verify(vk_bin, proofs[0], pubs[0]); print("single zkr_verify: %.2f ms" % (1e3 * (time.perf_counter() - t)))
