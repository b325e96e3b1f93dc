"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd SQLite output).
python profiles/summarize_pmc.py <fetch.db> <write.db> [out.json log_m round schedule]
schedule: "isolated" (ZKR_SERIAL=1: one stream, one proof at a time -- the default) or "pipelined" (the benchmarked schedule).
Units/corrections as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reads exactly half of a wide (16 B/lane) coalesced stream, so those reads are doubled; the gather kernels
take the factor measured for THEIR pattern (CALIBRATION below, round 6); WRITE_SIZE is uncalibrated.  The ingest_kernel
rows (known traffic: n*32 B read + n*32 B written, coalesced 16 B/lane) are the check of the stream factor."""
import re
import sqlite3
import sys


def load(path):
    db = sqlite3.connect(path)
    q = ("select s.kernel_name, p.value from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id")
    agg = {}
    for name, val in db.execute(q):
        m = re.search(r"L\d+([a-z_0-9]+kernel)", name)
        k = m.group(1) if m else name[:40]
        if "Fq2" in name:
            k += "<Fq2>"
        elif "FqParams" in name and ("msm_" in name or "fixed_base" in name or "gather" in name):
            k += "<Fq>"
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += val
    return agg


# FETCH_SIZE -> bytes requested from HBM, per ACCESS PATTERN (round 6).  The counter is calibrated against kernels
# of known traffic: tools/gather_bw (2^19 x 26 random gathers from an 832 MB table + their 4-byte indices, rocprofv3 --pmc FETCH_SIZE,
# profiles/r6_06_fetch_size_calibration.txt) and, for wide coalesced streams, the guide's x2 rule (MI355X_MICROARCH.md; check row:
# ingest_kernel, n * 32 B read).  factor = bytes requested / (FETCH_SIZE x 1024):
#   random 64-byte gathers   926.9 MB requested, FETCH_SIZE 1.135e6 KiB  -> 0.797   (G1 accumulation / oversized-bucket kernels)
#   random 128-byte gathers 1799.3 MB requested, FETCH_SIZE 1.106e6 KiB  -> 1.588   (G2 accumulation / oversized-bucket kernels)
#   wide coalesced streams                                               -> 2.0     (everything else)
CALIBRATION = {"gather64": 0.797, "gather128": 1.588, "stream": 2.0}
CALIBRATION_SOURCE = "profiles/r6_06_fetch_size_calibration.txt (tools/gather_bw under rocprofv3 --pmc FETCH_SIZE); streams: MI355X_MICROARCH.md x2 rule"


def pattern(kernel):
    if kernel.startswith("msm_accum") or kernel.startswith("msm_big_kernel"):
        return "gather128" if "<Fq2>" in kernel else "gather64"
    return "stream"


fetch, write = load(sys.argv[1]), load(sys.argv[2])
print("| kernel | launches | FETCH_SIZE KiB/launch | pattern (factor) | read MB/launch | WRITE_SIZE KiB/launch | write MB/launch | HBM MB/launch |")
print("|---|---|---|---|---|---|---|---|")
for k in sorted(fetch, key=lambda k: -(CALIBRATION[pattern(k)] * fetch[k][1] + write.get(k, [0, 0])[1])):
    n, f = fetch[k]
    w = write.get(k, [n, 0.0])[1]
    fr, wr = f / n, w / max(write.get(k, [n])[0], 1)
    c = CALIBRATION[pattern(k)]
    print("| `%s` | %d | %.0f | %s (%.3f) | %.2f | %.0f | %.2f | %.2f |" % (k, n, fr, pattern(k), c, c * fr * 1024 / 1e6, wr, wr * 1024 / 1e6, (c * fr + wr) * 1024 / 1e6))

if len(sys.argv) > 3:
    import json
    sched = sys.argv[6] if len(sys.argv) > 6 else "isolated"
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py "
                     + ("--steps 2 --warmup 1 --no-cpu-baseline --no-pipeline (ZKR_SERIAL=1)" if sched == "isolated" else "--steps 8 --warmup 2 --no-cpu-baseline (two proofs in flight, the benchmarked schedule)"),
           "schedule": "isolated kernels (ZKR_SERIAL=1)" if sched == "isolated" else "pipelined (two proofs in flight)",
           "correction": "KiB units; FETCH_SIZE x a factor per access pattern: " + ", ".join("%s %.3f" % kv for kv in sorted(CALIBRATION.items())) + "; WRITE_SIZE as counted",
           "calibration": CALIBRATION, "calibration_source": CALIBRATION_SOURCE,
           "config": {"log_m": int(sys.argv[4]), "n_public": 73}, "round": int(sys.argv[5]) if len(sys.argv) > 5 else 1, "kernels": {}}
    for k in fetch:
        n, f = fetch[k]
        wn, w = write.get(k, [n, 0.0])
        rb, wb = CALIBRATION[pattern(k)] * f / n * 1024, w / max(wn, 1) * 1024
        out["kernels"][k] = {"launches": n, "pattern": pattern(k), "fetch_size_raw_bytes_per_launch": f / n * 1024, "read_bytes_per_launch": rb, "write_bytes_per_launch": wb,
                             "hbm_bytes_per_launch": rb + wb}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
