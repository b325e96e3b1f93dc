"""Per-kernel average (per dispatch, summed over the counter's per-engine records) of one PMC counter from a rocprofv3 --pmc pass (rocpd SQLite): python profiles/summarize_counter.py <db> <name>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
q = ("select s.kernel_name, p.value, d.id from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.event_id "
     "join rocpd_info_kernel_symbol s on d.kernel_id = s.id")
agg = {}
for name, val, did in db.execute(q):
    m = re.search(r"L\d+([a-z_0-9]+kernel)", name)
    k = m.group(1) if m else name[:40]
    if "Fq2" in name:
        k += "<Fq2>"
    elif "FqParams" in name and ("msm_" in name or "fixed_base" in name or "gather" in name):
        k += "<Fq>"
    a = agg.setdefault(k, [set(), 0.0])     # a counter comes as one record per shader engine / XCD: sum them per dispatch
    a[0].add(did)
    a[1] += val
print("| kernel | launches | %s per launch |" % sys.argv[2])
print("|---|---|---|")
for k, (ids, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("| `%s` | %d | %.4g |" % (k, len(ids), v / len(ids)))
