"""Is the multiplier rate the bench line prices against a burst figure?  Runs the library's Fq-multiplication microbenchmark
(zkr_bench_fq_mul: 8 wavefronts per SIMD, registers only, best of three 30 ms launches per call) back to back for some seconds
and prints rate, sampled clock and board power per window, as G Fq-mul/s and as issue cycles per SIMD and second
(205 instructions x 4 cycles per product / 64 lanes / 1024 SIMDs) -- the unit DESIGN.md section 5 prices the proof in.
python tools/sustained_mul.py [seconds]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))
sys.path.insert(0, ROOT)
import zkr_hip
from bench import GpuSampler

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
zkr_hip.bench_fq_mul(0)
smp = GpuSampler(0, period=0.02).start()
t0 = time.perf_counter()
calls = []
while time.perf_counter() - t0 < seconds:
    a = time.perf_counter() - t0
    g = zkr_hip.bench_fq_mul(0)
    calls.append((a, time.perf_counter() - t0, g, len(smp.samples)))
state = smp.stop()
edges = [0, 0.5, 1, 2, 3, 4, 6, 8, 12, 1e9]
for lo, hi in zip(edges, edges[1:]):
    sel = [c for c in calls if lo <= c[0] < hi]
    if not sel:
        continue
    i0, i1 = sel[0][3], sel[-1][3] + 1
    mh = [m for m, _ in smp.samples[i0:i1 + 5] if m]
    pw = [w for _, w in smp.samples[i0:i1 + 5] if w]
    g = sum(c[2] for c in sel) / len(sel)
    print("%5.1f-%4.1f s: %6.1f G Fq-mul/s (best launch of each call; %d calls) = %.3f e9 issue cycles per SIMD and second; sclk %s MHz, power %s W" % (
        lo, min(hi, calls[-1][1]), g, len(sel), g * 205 * 4 / 64 / 1024, "%.0f" % (sum(mh) / len(mh)) if mh else "?", "%.0f" % (sum(pw) / len(pw)) if pw else "?"))
print("whole run:", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in state.items()})
