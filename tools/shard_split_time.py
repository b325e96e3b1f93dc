"""All shards of ONE proof concurrently on this GPU (zkr_prove_sharded_device): wall time per proof with calcH split over the shards
against every shard computing h for itself (ZKR_SHARD_SPLIT_H=0), and the split's phase times per shard.  On one GPU the shards
share the chip, so the wall time is the AGGREGATE work of the proof, not a latency: what it shows is the work a split saves.
python tools/shard_split_time.py [log_m=22] [parts=8] [n=5]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "simple-zk-rollups_amd", "python"))
import torch
import zkr_hip

log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 22
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
key, w0, _ = zkr_hip.ProvingKey.synth(log_m, device=0, want_aux=False)
w = torch.frombuffer(bytearray(w0), dtype=torch.uint8).cuda(0)
torch.cuda.synchronize()
whole = key.prove_device(w.data_ptr(), r=3, s=4)
t = time.perf_counter()
for _ in range(n):
    key.prove_device(w.data_ptr(), r=3, s=4)
whole_ms = 1e3 * (time.perf_counter() - t) / n
shards = [key.shard(i, parts, 0) for i in range(parts)]
key.close()
ptrs = [w.data_ptr()] * parts
for mode in ("1", "0"):
    os.environ["ZKR_SHARD_SPLIT_H"] = mode
    for _ in range(2):
        assert zkr_hip.prove_sharded_device(shards, ptrs, 3, 4) == whole
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        zkr_hip.prove_sharded_device(shards, ptrs, 3, 4)
        ts.append(1e3 * (time.perf_counter() - t))
    ts.sort()
    stats = zkr_hip.sharded_split_stats()
    print("2^%d, %d shards on one GPU, split_h=%s: median %.2f ms per proof (min %.2f, max %.2f); whole key, synchronous: %.2f ms" % (log_m, parts, mode, ts[len(ts) // 2], ts[0], ts[-1], whole_ms))
    if stats:
        print("   phases (QAP rows, cross 1, blocks, cross 2, tail enqueue), ms, max over shards: " + ", ".join("%.3f" % max(r[f] for r in stats) for f in range(5)))
        print("   sum of the per-phase maxima: %.3f ms" % sum(max(r[f] for r in stats) for f in range(5)))
