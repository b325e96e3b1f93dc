"""Soak test of the concurrent paths on one GPU: for `seconds`, several host threads prove on the SAME key material at once --
batches over two replicas (zkr_prove_batch_multi), sharded proofs over four shards from two callers (zkr_prove_sharded), single synchronous
proofs and pipelined device batches on the whole key -- with random blinding; every proof goes through the native verifier
(zkr_verify_batch) and a sample is compared with the toxic-waste closed form (the oracle: that is why this driver lives under
tests/, not tools/).  Prints counts; exits non-zero on any failure.
    python tests/soak.py [log_m=16] [seconds=120]"""
import os
import sys
import threading
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import groth16 as g  # noqa: E402  (the checker)
import torch  # noqa: E402
import zkr_hip  # noqa: E402

log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
P = 73
key, w0, aux = zkr_hip.ProvingKey.synth(log_m, P, 0x5A4B0001, 0x5A4B00FF)
vk_bin = key.synth_vk(aux)
wbs = [w0] + [zkr_hip.synth_witness(log_m, P, 0x5A4B0001, 6000 + i) for i in range(1, 5)]
pubs = [[int.from_bytes(w[32 * j:32 * j + 32], "little") for j in range(1, P + 1)] for w in wbs]
dws = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda(0) for w in wbs]
torch.cuda.synchronize()
rep = key.replicate(0)
shards = [key.shard(i, 4, 0) for i in range(4)]
stop = time.time() + seconds
counts, failures, lock = {}, [], threading.Lock()


def note(tag, n, ok, what=""):
    with lock:
        counts[tag] = counts.get(tag, 0) + n
        if not ok:
            failures.append((tag, what))


def multi_batches(tag, seed):
    i = seed
    while time.time() < stop and not failures:
        n = 3 + i % 9
        idx = [(i + j) % len(wbs) for j in range(n)]
        proofs = zkr_hip.prove_batch_multi([key, rep], [wbs[k] for k in idx])
        note(tag, n, zkr_hip.verify_batch(vk_bin, proofs, [pubs[k] for k in idx]) and len(set(proofs)) == n, "batch %d" % i)
        i += 1


def sharded(tag, start=0):
    i = start
    while time.time() < stop and not failures:
        k = i % len(wbs)
        r, s = 1000 + i, 7000 + 3 * i
        proof = zkr_hip.prove_sharded(shards, wbs[k], r, s) if i % 2 else zkr_hip.prove_sharded_device(shards, [dws[k].data_ptr()] * 4, r, s)
        ok = zkr_hip.verify(vk_bin, proof, pubs[k])
        if ok and i % 16 == 0:
            ok = proof == g.proof_bytes(g.proof_from_aux(aux, wbs[k], P, r, s)[0])
        note(tag, 1, ok, "sharded %d" % i)
        i += 1


def singles(tag):
    i = 0
    while time.time() < stop and not failures:
        k = i % len(wbs)
        proof = key.prove(wbs[k]) if i % 3 else key.prove_device(dws[k].data_ptr())
        note(tag, 1, zkr_hip.verify(vk_bin, proof, pubs[k]), "single %d" % i)
        i += 1


def device_batches(tag):
    i = 0
    while time.time() < stop and not failures:
        n = 2 + i % 6
        idx = [(i + 2 * j) % len(wbs) for j in range(n)]
        proofs = rep.prove_batch_device([dws[k].data_ptr() for k in idx])
        note(tag, n, zkr_hip.verify_batch(vk_bin, proofs, [pubs[k] for k in idx]), "device batch %d" % i)
        i += 1


threads = [threading.Thread(target=multi_batches, args=("multi_a", 0)), threading.Thread(target=multi_batches, args=("multi_b", 5)),
           threading.Thread(target=sharded, args=("sharded",)), threading.Thread(target=sharded, args=("sharded_b", 1)),  # two at once: they take turns (calcH split over the shards)
           threading.Thread(target=singles, args=("single",)),
           threading.Thread(target=device_batches, args=("device_batch",))]
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info(0)[0]
import resource
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
samples = []
def sampler():
    while time.time() < stop and not failures:
        time.sleep(min(15.0, max(0.1, stop - time.time())))
        samples.append(torch.cuda.mem_get_info(0)[0])
threads.append(threading.Thread(target=sampler))
t0 = time.time()
for t in threads:
    t.start()
for t in threads:
    t.join()
el = time.time() - t0
total = sum(counts.values())
print("soak 2^%d, %.0f s, 6 host threads on one key + one replica + four shards: %d proofs verified (%s), %.0f proofs/s, failures: %s"
      % (log_m, el, total, ", ".join("%s %d" % kv for kv in sorted(counts.items())), total / el, failures or "none"))
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info(0)[0]
rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("device memory free before / after: %.1f / %.1f MiB (difference %.1f MiB); host peak RSS before / after: %.0f / %.0f MiB"
      % (free0 / 2 ** 20, free1 / 2 ** 20, (free0 - free1) / 2 ** 20, rss0 / 1024, rss1 / 1024))
print("device memory in use beyond the start, every 15 s (MiB): " + ", ".join("%.0f" % ((free0 - f) / 2 ** 20) for f in samples) + "  (lazily built workspaces, then flat: no leak)")
if len(samples) >= 3 and samples[-1] < samples[1] - (64 << 20):
    failures.append(("memory", "device memory kept growing: %s" % samples))
    sys.exit(1)
sys.exit(1 if failures else 0)
