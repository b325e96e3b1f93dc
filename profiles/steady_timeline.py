"""One steady-state proof period of the pipelined bench, kernel by kernel, from a rocprofv3 kernel trace (rocpd SQLite):
python profiles/steady_timeline.py <results.db> [which]
The window runs from the start of one G2 accumulation (the first launch of a proof's accumulations) to the start of the next one,
taken in the middle of the run (`which`: offset from the middle, default 0).  Every dispatch that overlaps the window is listed
with its queue, start and end relative to the window's start (ms) and duration (us); then per queue the busy time inside the window."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select s.kernel_name, d.start, d.end, d.queue_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                  "on d.kernel_id = s.id order by d.start").fetchall()


def short(n):
    m = re.search(r"zkrL\d+([a-z_0-9]+?)(?:I|E)", n)
    base = m.group(1) if m else n[:24]
    return base + ("<Fq2>" if "Fq2" in n else "<Fq>" if "FqParams" in n and "msm" in n else "")


g2 = [r[1] for r in rows if "msm_accum_kernel" in r[0] and "Fq2" in r[0]]
k = len(g2) // 2 + (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, t1 = g2[k], g2[k + 1]
print("window: %.3f ms between two G2 accumulation starts (proof period)" % ((t1 - t0) / 1e6))
qs = sorted({r[3] for r in rows if r[2] > t0 and r[1] < t1})
busy = {q: 0 for q in qs}
print("| queue | kernel | start ms | end ms | us |")
print("|---|---|---|---|---|")
for n, st, en, q in rows:
    if en <= t0 or st >= t1:
        continue
    busy[q] += min(en, t1) - max(st, t0)
    print("| %s | %s | %.3f | %.3f | %.1f |" % (q, short(n), (st - t0) / 1e6, (en - t0) / 1e6, (en - st) / 1e3))
print()
for q in qs:
    print("queue %s busy %.3f ms of %.3f" % (q, busy[q] / 1e6, (t1 - t0) / 1e6))
