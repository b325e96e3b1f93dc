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

# ---- round 5: what the host-buffer call adds (is there anything for a chunked upload to hide?)
# (1) the same proof from a pageable host buffer (zkr_prove: upload inside the call); (2) the bare upload: one pageable
# host-to-device copy of the witness, nothing else running; (3) the kernels a chunked upload could run behind the copy
# (ingest + the digit counts of w), from the key's own stage timers.
hw = bytes(w0)
for _ in range(3):
    key.prove(hw, 5, 7)
th = []
for i in range(n):
    t = time.perf_counter()
    key.prove(hw, 11 + i, 13 + i)
    th.append(1e3 * (time.perf_counter() - t))
th.sort()
hsrc = torch.frombuffer(bytearray(w0), dtype=torch.uint8)          # pageable
dst = torch.empty_like(w)
tc = []
for i in range(n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    dst.copy_(hsrc)
    torch.cuda.synchronize()
    tc.append(1e3 * (time.perf_counter() - t))
tc.sort()
key.prof_enable(True)
key.prof_reset()
for i in range(4):
    key.prove_device(w.data_ptr(), r=11 + i, s=13 + i, stream=stream)
pr = key.prof()
key.prof_enable(False)
hide = sum(pr[k][0] for k in ("ingest",) if k in pr) / 4
print("host-buffer 2^%d proof: median %.3f ms (resident %.3f: +%.3f); bare pageable upload of %.1f MB: median %.3f ms (%.1f GB/s); "
      "ingest alone %.3f ms per proof -- all a chunked upload could overlap besides the first digit kernel (0.05 ms)"
      % (log_m, th[len(th) // 2], ts[len(ts) // 2], th[len(th) // 2] - ts[len(ts) // 2], len(hw) / 1e6, tc[len(tc) // 2], len(hw) / tc[len(tc) // 2] / 1e6, hide))
