"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd SQLite output).
python profiles/summarize_pmc.py <fetch.db> <write.db> [out.json log_m round schedule]
schedule: "isolated" (ZKR_SERIAL=1: one stream, one proof at a time -- the default) or "pipelined" (the benchmarked schedule).
Units/corrections as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reads exactly half of a wide (16 B/lane) coalesced stream, so reads are doubled; WRITE_SIZE is
uncalibrated.  The ingest_kernel rows (known traffic: n*32 B read + n*32 B written, coalesced 16 B/lane)
are printed as the calibration check for this access width."""
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


fetch, write = load(sys.argv[1]), load(sys.argv[2])
print("| kernel | launches | FETCH_SIZE KiB/launch | read MB/launch (x2 gfx950) | WRITE_SIZE KiB/launch | write MB/launch | HBM MB/launch |")
print("|---|---|---|---|---|---|---|")
for k in sorted(fetch, key=lambda k: -(2 * fetch[k][1] + write.get(k, [0, 0])[1])):
    n, f = fetch[k]
    w = write.get(k, [n, 0.0])[1]
    fr, wr = f / n, w / max(write.get(k, [n])[0], 1)
    print("| `%s` | %d | %.0f | %.2f | %.0f | %.2f | %.2f |" % (k, n, fr, 2 * fr * 1024 / 1e6, wr, wr * 1024 / 1e6, (2 * fr + wr) * 1024 / 1e6))

if len(sys.argv) > 3:
    import json
    sched = sys.argv[6] if len(sys.argv) > 6 else "isolated"
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py "
                     + ("--steps 2 --warmup 1 --no-cpu-baseline --no-pipeline (ZKR_SERIAL=1)" if sched == "isolated" else "--steps 8 --warmup 2 --no-cpu-baseline (two proofs in flight, the benchmarked schedule)"),
           "schedule": "isolated kernels (ZKR_SERIAL=1)" if sched == "isolated" else "pipelined (two proofs in flight)",
           "correction": "KiB units; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md HBM section); calibration row: ingest_kernel (n*32 B read + n*32 B written)",
           "config": {"log_m": int(sys.argv[4]), "n_public": 73}, "round": int(sys.argv[5]) if len(sys.argv) > 5 else 1, "kernels": {}}
    for k in fetch:
        n, f = fetch[k]
        wn, w = write.get(k, [n, 0.0])
        rb, wb = 2 * f / n * 1024, w / max(wn, 1) * 1024
        out["kernels"][k] = {"launches": n, "read_bytes_per_launch": rb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": rb + wb}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
