"""One synchronous proof at a time of the synthetic rollup circuit (bench.py's workload): latency per call, nothing else running.
python tools/sync_single.py [log_m=20] [n=12]   (under rocprofv3 --kernel-trace for the timeline of one proof: profiles/timeline.py <db> 3)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "simple-zk-rollups_amd", "python"))
import torch
import zkr_hip

log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
key, w0, _ = zkr_hip.ProvingKey.synth(log_m, device=0, want_aux=False)
w = torch.frombuffer(bytearray(w0), dtype=torch.uint8).cuda(0)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    key.prove_device(w.data_ptr(), r=5, s=7, stream=stream)
ts = []
for i in range(n):
    t = time.perf_counter()
    key.prove_device(w.data_ptr(), r=11 + i, s=13 + i, stream=stream)
    ts.append(1e3 * (time.perf_counter() - t))
ts.sort()
print("synchronous 2^%d proof: median %.3f ms, min %.3f, max %.3f over %d calls" % (log_m, ts[len(ts) // 2], ts[0], ts[-1], n))
