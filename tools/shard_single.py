"""One shard's share of one proof, synchronously (zkr_prove_partial_device): latency per call, nothing else running.
python tools/shard_single.py [log_m=22] [parts=8] [part=3] [n=6] [split=0]   (under rocprofv3 --kernel-trace: profiles/timeline.py <db> 2)
split=1: the share with calcH split over the shards, own buffers standing in for the others' (zkr_bench_shard_split_solo)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "simple-zk-rollups_amd", "python"))
import torch
import zkr_hip

log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 22
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
part = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = int(sys.argv[4]) if len(sys.argv) > 4 else 6
split = len(sys.argv) > 5 and sys.argv[5] == "1"
key, w0, _ = zkr_hip.ProvingKey.synth(log_m, device=0, want_aux=False)
sh = key.shard(part, parts, 0)
key.close()
w = torch.frombuffer(bytearray(w0), dtype=torch.uint8).cuda(0)
torch.cuda.synchronize()
run = (lambda: sh.bench_split_solo(w.data_ptr())) if split else (lambda: sh.prove_partial_device(w.data_ptr()))
for _ in range(2):
    run()
ts = []
for i in range(n):
    t = time.perf_counter()
    run()
    ts.append(1e3 * (time.perf_counter() - t))
ts.sort()
print("shard %d of %d at 2^%d%s: median %.3f ms, min %.3f, max %.3f over %d calls" % (part, parts, log_m, ", calcH split" if split else "", ts[len(ts) // 2], ts[0], ts[-1], n))
sh.prof_enable(True)
sh.prof_reset()
run()
print("   stage sums (ms): " + ", ".join("%s %.3f" % (k, v[0]) for k, v in sh.prof().items() if v[1]))
