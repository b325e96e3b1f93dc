"""How busy each hardware queue is, and which kernels overlap the accumulations, from a rocprofv3 kernel trace (rocpd SQLite) of the
pipelined bench: python profiles/occupancy_timeline.py <results.db>
Per queue: busy fraction of the steady-state window; for the accumulation kernels: time spent alone / together with kernels of other queues."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select s.kernel_name, d.start, d.end, d.queue_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                  "on d.kernel_id = s.id order by d.start").fetchall()
ing = [r[1] for r in rows if "ingest_kernel" in r[0]]
t0, t1 = ing[len(ing) // 4], ing[-2]          # steady state: skip key build / warm-up and the tail
win = (t1 - t0) / 1e6
rows = [r for r in rows if r[1] >= t0 and r[2] <= t1]


def short(n):
    m = re.search(r"zkrL\d+([a-z_0-9]+?)(?:I|E)", n)
    base = m.group(1) if m else n[:24]
    return base + ("<Fq2>" if "Fq2" in n else "<Fq>" if "FqParams" in n and "msm" in n else "")


proofs = sum(1 for r in rows if "ingest_kernel" in r[0])
print("window %.1f ms, %d proofs -> %.3f ms per proof" % (win, proofs, win / max(proofs, 1)))
byq = {}
for n, st, en, q in rows:
    byq.setdefault(q, []).append((st, en, short(n)))
for q, lst in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    lst.sort()
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in lst:                       # union of intervals (a queue runs one kernel at a time, but be safe)
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    names = {}
    for s, e, n in lst:
        names[n] = names.get(n, 0) + (e - s)
    top = ", ".join("%s %.2f" % (k, v / 1e6 / max(proofs, 1)) for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:5])
    print("queue %s: busy %.1f %% of the window (%.2f ms per proof): %s" % (q, 100 * busy / 1e6 / win, busy / 1e6 / max(proofs, 1), top))
# the accumulation kernels against everything else: sample the timeline
ev = []
for n, st, en, q in rows:
    kind = "acc" if "msm_accum" in n else "other"
    ev.append((st, 1, kind)); ev.append((en, -1, kind))
ev.sort()
cnt = {"acc": 0, "other": 0}
last = t0
tim = {}
for t, d, kind in ev:
    key = ("acc" if cnt["acc"] else "-") + "+" + ("%d other" % min(cnt["other"], 3) if cnt["other"] else "nothing else")
    tim[key] = tim.get(key, 0) + (t - last)
    last = t
    cnt[kind] += d
for k, v in sorted(tim.items(), key=lambda kv: -kv[1]):
    print("%-28s %.2f ms per proof (%.1f %%)" % (k, v / 1e6 / max(proofs, 1), 100 * v / 1e6 / win))
