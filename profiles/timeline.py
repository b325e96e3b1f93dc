"""Per-proof kernel timeline from a rocprofv3 rocpd SQLite kernel trace (concurrent streams):
python profiles/timeline.py <results.db> [proof_index_from_end=1]
Prints start/end (ms, relative to the proof's ingest_kernel) of every dispatch of one proof."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = db.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x, d.workgroup_size_x "
                  "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
starts = [i for i, r in enumerate(rows) if "ingest_kernel" in r[0]]
i0 = starts[-back]
i1 = starts[-back + 1] if back > 1 else len(rows)
t0 = rows[i0][1]


def short(n):
    n = re.sub(r"\s*\[clone .*\]$", "", n)
    m = re.search(r"zkrL\d+([a-z_0-9]+?)(?:I|E)", n)
    base = m.group(1) if m else n[:30]
    tag = "<Fq2>" if "Fq2" in n else ("<Fq>" if "FqParams" in n else "")
    return base + tag


queues = {}
for name, st, en, q, gx, wx in rows[i0:i1]:
    qi = queues.setdefault(q, len(queues))
    if "bench" in name:
        continue
    print("%8.3f %8.3f  %7.3f  q%d  %-32s grid %d" % ((st - t0) / 1e6, (en - t0) / 1e6, (en - st) / 1e6, qi, short(name), gx // max(wx, 1)))
