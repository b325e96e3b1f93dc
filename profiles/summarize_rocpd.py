"""Summarise a rocprofv3 rocpd SQLite result (kernel trace) into a per-kernel stats table:
python profiles/summarize_rocpd.py <results.db> [skip_first_dispatches_of_each_kernel]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.group_segment_size, d.private_segment_size "
                  "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
agg = {}
for name, st, en, gx, wx, lds, scr in rows:
    name = re.sub(r"\s*\[clone .*\]$", "", name)
    a = agg.setdefault(name, dict(n=0, tot=0, mn=1e30, mx=0, grid=gx, wg=wx, lds=lds, scratch=scr))
    dur = en - st
    a["n"] += 1; a["tot"] += dur; a["mn"] = min(a["mn"], dur); a["mx"] = max(a["mx"], dur)
total = sum(a["tot"] for a in agg.values())
print("| kernel | calls | total ms | avg us | min us | max us | % | grid | wg | lds B | scratch B |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
    short = name if len(name) < 110 else name[:107] + "..."
    print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %d | %d | %d | %d |" % (short, a["n"], a["tot"] / 1e6, a["tot"] / a["n"] / 1e3, a["mn"] / 1e3,
          a["mx"] / 1e3, 100.0 * a["tot"] / total, a["grid"], a["wg"], a["lds"] or 0, a["scratch"] or 0))
print("\ntotal kernel time %.3f ms over %d dispatches" % (total / 1e6, len(rows)))
