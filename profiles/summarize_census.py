"""VALU census of one proof from a rocprofv3 --pmc pass that carried several SQ counters at once (rocpd SQLite).

python profiles/summarize_census.py <db> <launches_per_proof.json|auto|-> [step_ms sclk_ghz [out.json log_m round]]   (auto: launches / number of G2 accumulations)

Per kernel (per launch, counters summed over their per-XCD / per-SE records): duration, SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU,
SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY and the derived
  valu_busy  = 4 * SQ_ACTIVE_INST_VALU / (SIMDS * clk * duration)   (SQ_ACTIVE_INST_* count quad-cycles, MI355X_MICROARCH.md)
  cyc/inst   = 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU              (issue cycles one wave64 VALU instruction holds its SIMD)
where clk = GRBM_GUI_ACTIVE / duration of the same launch when that counter is in the pass (else the given sclk).
With step_ms: the proof's VALU-busy fraction = sum over kernels(launches per proof x 4 x SQ_ACTIVE_INST_VALU) / (SIMDS x sclk x step).
"""
import json
import re
import sqlite3
import sys

SIMDS = 1024  # 256 CUs x 4 SIMDs

db = sqlite3.connect(sys.argv[1])
q = ("select s.kernel_name, i.name, p.value, d.id, d.start, d.end from rocpd_pmc_event p "
     "join rocpd_info_pmc i on p.pmc_id = i.id join rocpd_kernel_dispatch d on p.event_id = d.event_id "
     "join rocpd_info_kernel_symbol s on d.kernel_id = s.id")
agg = {}
counters = set()
for name, cname, val, did, t0, t1 in db.execute(q):
    m = re.search(r"L\d+([a-z_0-9]+kernel)", name)
    k = m.group(1) if m else name[:40]
    if "Fq2" in name:
        k += "<Fq2>"
    elif "FqParams" in name and ("msm_" in name or "fixed_base" in name or "gather" in name):
        k += "<Fq>"
    a = agg.setdefault(k, {"ids": {}, "c": {}})
    a["ids"][did] = (t1 - t0)
    a["c"][cname] = a["c"].get(cname, 0.0) + val
    counters.add(cname)

per_proof = None
if len(sys.argv) > 2 and sys.argv[2] == "auto":
    # one G2 accumulation per proof; kernels of key build / setup / microbenchmarks are not part of a proof
    proofs = len(agg["msm_accum_kernel<Fq2>"]["ids"])
    skip = ("precompute", "fixed_base", "bench", "twiddle", "gather_kernel", "rocclr", "radix_convert", "f29_forms", "workload", "synth", "r1cs", "setup")
    per_proof = {k: len(a["ids"]) / proofs for k, a in agg.items() if not any(x in k for x in skip)}
elif len(sys.argv) > 2 and sys.argv[2] != "-":
    per_proof = json.load(open(sys.argv[2]))
step_ms = float(sys.argv[3]) if len(sys.argv) > 3 else None
sclk = float(sys.argv[4]) * 1e9 if len(sys.argv) > 4 else 2.3e9

cols = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"]
cols = [c for c in cols if c in counters] + sorted(counters - set(cols))
print("counters in this pass:", ", ".join(cols))
print()
print("| kernel | launches | avg us | " + " | ".join(cols) + " | clk GHz | VALU busy | cyc / VALU inst | wait_inst / wave_cyc | wait_any / wave_cyc |")
print("|---" * (9 + len(cols)) + "|")
rows = []
for k, a in agg.items():
    n = len(a["ids"])
    dur = sum(a["ids"].values()) / n * 1e-9  # s
    c = {x: a["c"].get(x, 0.0) / n for x in cols}
    # GRBM_GUI_ACTIVE comes as one record per XCD/SE instance: its per-launch SUM / instances would be the clock; take it from the ratio to SQ_BUSY_CYCLES when present
    rows.append((k, n, dur, c))
tot_active = 0.0
tot_insts = 0.0
for k, n, dur, c in sorted(rows, key=lambda r: -r[3].get("SQ_ACTIVE_INST_VALU", 0) * r[1]):
    act = c.get("SQ_ACTIVE_INST_VALU", 0.0)
    ins = c.get("SQ_INSTS_VALU", 0.0)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    busy = 4 * act / (SIMDS * sclk * dur) if dur > 0 else 0
    print("| `%s` | %d | %.1f | %s | %s | %.3f | %.2f | %.3f | %.3f |" % (
        k, n, dur * 1e6, " | ".join("%.4g" % c[x] for x in cols), "-", busy, 4 * act / ins if ins else 0,
        c.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0, c.get("SQ_WAIT_ANY", 0) / wc if wc else 0))
    if per_proof and k in per_proof:
        tot_active += per_proof[k] * act
        tot_insts += per_proof[k] * ins
if per_proof:
    print()
    print("launches per proof: " + ", ".join("%s %.2f" % (k, v) for k, v in sorted(per_proof.items())))
    print("per proof (launch counts from %s): SQ_ACTIVE_INST_VALU %.4g quad-cycles = %.4g SIMD-cycles, SQ_INSTS_VALU %.4g" % (sys.argv[2], tot_active, 4 * tot_active, tot_insts))
    if step_ms:
        print("VALU-busy fraction of a %.3f ms step at %.2f GHz over %d SIMDs: %.3f" % (step_ms, sclk / 1e9, SIMDS, 4 * tot_active / (SIMDS * sclk * step_ms * 1e-3)))
        print("step time at 100 %% VALU-busy: %.3f ms" % (4 * tot_active / (SIMDS * sclk) * 1e3))

if per_proof and len(sys.argv) > 5:
    out = {"source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ... GRBM_GUI_ACTIVE (one pass) -- ZKR_SERIAL=1 python3 bench.py --steps 2 --warmup 1 --no-pipeline (isolated kernels)",
           "config": {"log_m": int(sys.argv[6]), "n_public": 73}, "round": int(sys.argv[7]),
           "valu_wave_instructions_per_proof": tot_insts, "valu_active_quad_cycles_per_proof": tot_active, "launches_per_proof": per_proof, "kernels": {}}
    for k, n, dur, c in rows:
        if k in per_proof:
            out["kernels"][k] = {"launches": n, "avg_us": dur * 1e6, "SQ_INSTS_VALU_per_launch": c.get("SQ_INSTS_VALU", 0.0), "SQ_WAVE_CYCLES_per_launch": c.get("SQ_WAVE_CYCLES", 0.0),
                                "SQ_WAIT_ANY_per_launch": c.get("SQ_WAIT_ANY", 0.0), "SQ_WAIT_INST_ANY_per_launch": c.get("SQ_WAIT_INST_ANY", 0.0), "GRBM_GUI_ACTIVE_per_launch": c.get("GRBM_GUI_ACTIVE", 0.0)}
    json.dump(out, open(sys.argv[5], "w"), indent=1)
