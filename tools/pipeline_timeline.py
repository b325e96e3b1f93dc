"""Timeline of the facade pipeline's stages (GPU witnesses): per chunk when its witnesses were ready, when its proofs started
and ended, when its verification ended -- to see which stage the end-to-end rate waits for."""
import os, sys, time, threading, queue
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))
import torch, zkr_hip
from zkr_hip import rollup
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 32
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
w = circ.calculate_witness_batch_device(flats[:8])
key.prove_batch_device([w[i].data_ptr() for i in range(8)])
# ceiling: all witnesses first, then prove calls of `chunk`, verification on a thread
wall = circ.calculate_witness_batch_device(flats)
torch.cuda.synchronize()
for csz in (32, 64, 128, 256):
    t0 = time.perf_counter()
    for c0 in range(0, n, csz):
        key.prove_batch_device([wall[i].data_ptr() for i in range(c0, min(n, c0 + csz))])
    el = time.perf_counter() - t0
    print("prove only, calls of %3d: %.0f proofs/s" % (csz, n / el))
ev = []
q_wit, q_ver = queue.Queue(maxsize=2), queue.Queue()
T0 = time.perf_counter()
now = lambda: 1e3 * (time.perf_counter() - T0)
def wthread():
    for ci, c0 in enumerate(range(0, n, chunk)):
        a = now(); t = circ.calculate_witness_batch_device(flats[c0:c0 + chunk]); ev.append(("W%d" % ci, a, now())); q_wit.put(t)
    q_wit.put(None)
def vthread():
    ci = 0
    while True:
        it = q_ver.get()
        if it is None: return
        a = now(); ok = zkr_hip.verify_batch(vk_bin, it[0], it[1]); ev.append(("V%d" % ci, a, now())); ci += 1
wt, vt = threading.Thread(target=wthread), threading.Thread(target=vthread)
wt.start(); vt.start()
ci = 0
while True:
    t = q_wit.get()
    if t is None: break
    a = now(); proofs = key.prove_batch_device([t[i].data_ptr() for i in range(t.shape[0])]); b = now()
    head = t[:, 32:32 * (circ.n_public + 1)].cpu().numpy()
    pubs = [[int.from_bytes(row[32 * j:32 * j + 32].tobytes(), "little") for j in range(circ.n_public)] for row in head]
    ev.append(("P%d" % ci, a, b)); ev.append(("pub%d" % ci, b, now())); ci += 1
    q_ver.put((proofs, pubs))
q_ver.put(None); wt.join(); vt.join()
print("end to end: %.1f ms -> %.0f proofs/s" % (now(), n / (now() * 1e-3)))
for name, a, b in sorted(ev, key=lambda e: e[1]):
    print("%-6s %7.1f -> %7.1f  (%.1f ms)" % (name, a, b, b - a))
