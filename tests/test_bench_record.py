"""bench.py's line as the DRIVER keeps it: the driver's record holds the scalars and the first ~120 characters of the
strings of `config`, `roofline` and `cpu_baseline`, with key names cut at 40 characters -- so the figures that matter are scalars
with short names next to the nested objects.  CPU only: the helpers are exercised on recorded legs (profiles/r5_*), no GPU."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _driver_view(obj):
    """What the driver keeps of one of the three objects: scalars, strings cut at 120, key names cut at 40."""
    return {k[:40]: (v[:120] if isinstance(v, str) else v) for k, v in obj.items() if not isinstance(v, (dict, list))}


def test_recorded_default_line_carries_the_claim_as_scalars():
    d = json.load(open(os.path.join(ROOT, "profiles", "r5_09_bench_default.json")))
    cfg, roof, cpu = _driver_view(d["config"]), _driver_view(d["roofline"]), _driver_view(d["cpu_baseline"])
    # no two keys collide once cut at 40 characters
    for obj in (d["config"], d["roofline"], d["cpu_baseline"]):
        flat = [k for k, v in obj.items() if not isinstance(v, (dict, list))]
        assert len({k[:40] for k in flat}) == len(flat)
    # the reference's calling pattern, the other BASELINE config, the reference's two circuits, the sharding leg
    for k in ("sync_latency_ms", "host_buffer_sync_proofs_per_s", "host_buffer_batch_proofs_per_s", "rate_2_22_proofs_per_s", "ms_per_proof_2_22",
              "tx_single_proof_ms", "tx_fused_proofs_per_s", "tx_dropin_call_ms", "withdraw_single_proof_ms", "withdraw_fused_proofs_per_s",
              "withdraw_dropin_call_ms", "sharded_parts", "sharded_ms", "sharded_speedup", "sharded_measured", "sharded_form"):
        assert cfg.get(k) is not None, k
    assert cfg["sharded_measured"] is False and cfg["sharded_form"].startswith("projected")        # one GPU: a projection, and it says so
    for k in ("frac", "achieved", "peak", "traffic", "valu_frac_kernel", "valu_frac_proof", "mad_bound_frac_proof", "hbm_traffic_frac_proof", "frac_2_22",
              "hbm_frac_ingest", "hbm_frac_spmv", "hbm_frac_ntt_pass", "hbm_frac_combine_h"):
        assert isinstance(roof.get(k), float), k
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and 0 < roof["frac_2_22"] < 0.05
    for k in ("value", "cores", "all_threads_proofs_per_s", "all_threads_cores", "seconds_per_proof"):
        assert cpu.get(k) is not None, k
    assert cpu["kind"] == "port" and cpu["cpu_and_gpu_proofs_identical"] is True
    assert len(d["cpu_baseline"]["sample"]) <= 120 and "MEASURED" in cpu["sample"] and "march=" in cpu["sample"]   # survives the cut whole
    # what the nested objects say and what the scalars say is the same number
    assert cfg["sync_latency_ms"] == d["config"]["boundary"]["sync_latency_ms"]
    assert cfg["rate_2_22_proofs_per_s"] == d["config_2_22"]["proofs_per_s"] and d["config_2_22"]["proofs_verified"] == d["config_2_22"]["proofs"]
    assert cfg["withdraw_single_proof_ms"] == d["withdraw_circuit"]["single_proof_ms"]
    assert roof["valu_frac_proof"] == d["roofline"]["valu"]["whole_proof"]["frac"]


def test_sharding_scalars_have_one_schema_for_projected_measured_and_failed_legs():
    b = _bench()
    proj = json.load(open(os.path.join(ROOT, "profiles", "r5_09_bench_default.json")))["intra_proof_sharding"]
    meas = json.load(open(os.path.join(ROOT, "profiles", "r5_08_bench_inproc8_one_gpu.json")))["intra_proof_sharding"]
    a, m = b.sharding_scalars(proj), b.sharding_scalars(meas)
    common = {"sharded_parts", "sharded_measured", "sharded_form", "sharded_ms", "sharded_speedup", "sharded_replicated_calch_ms", "sharded_whole_key_ms"}
    assert common <= set(a) and common <= set(m)
    assert a["sharded_measured"] is False and m["sharded_measured"] is False          # eight shards on ONE device: a rehearsal is not a measurement
    assert m["sharded_form"] == "split" and m["sharded_parts"] == 8 and "one device" in m["sharded_reason"]
    real = dict(meas, rehearsal_on_one_gpu=False)                                      # the same leg as a multi-GPU node would report it
    assert b.sharding_scalars(real)["sharded_measured"] is True
    assert b.sharding_scalars({"error": "boom"}) == {"sharded_error": "boom"} and b.sharding_scalars(None) == {"sharded_error": None}


def test_north_star_ratios_are_scalars_of_cpu_baseline():
    """the denominator of the ">= 50x snarkjs single-thread" claim and the three ratios are scalars (the driver's
    record drops the nested `snarkjs_style` object)."""
    b = _bench()
    d = json.load(open(os.path.join(ROOT, "profiles", "r5_09_bench_default.json")))
    cb = dict(d["cpu_baseline"])
    b.cpu_baseline_scalars(cb, d["value"], 12)
    cpu = _driver_view(cb)
    for k in ("snarkjs_style_proofs_per_s", "snarkjs_style_sample_log_m", "snarkjs_style_seconds_per_proof", "speedup_vs_snarkjs_style",
              "speedup_vs_c_1thread", "speedup_vs_c_all_threads", "cores", "all_threads_cores"):
        assert isinstance(cpu.get(k), (int, float)), k
    assert cpu["snarkjs_style_proofs_per_s"] == d["cpu_baseline"]["snarkjs_style"]["value"] and cpu["snarkjs_style_sample_log_m"] == 12
    assert abs(cpu["speedup_vs_snarkjs_style"] * cpu["snarkjs_style_proofs_per_s"] - d["value"]) < 1e-9
    assert abs(cpu["speedup_vs_c_1thread"] * cpu["value"] - d["value"]) < 1e-9
    assert abs(cpu["speedup_vs_c_all_threads"] * cpu["all_threads_proofs_per_s"] - d["value"]) < 1e-9
    assert cpu["speedup_vs_snarkjs_style"] >= 50                                        # the north star's single-GPU target
    flat = [k for k, v in cb.items() if not isinstance(v, (dict, list))]
    assert len({k[:40] for k in flat}) == len(flat)
    # no JS leg (node missing / --no-js-baseline): the keys exist and are null, the C ratios stay
    cb2 = {k: v for k, v in d["cpu_baseline"].items() if k != "snarkjs_style"}
    b.cpu_baseline_scalars(cb2, d["value"], 12)
    assert cb2["snarkjs_style_proofs_per_s"] is None and cb2["speedup_vs_snarkjs_style"] is None and cb2["speedup_vs_c_1thread"] > 1


def test_multi_rank_line_schema_and_what_counts_as_measured():
    """a `--gpus N` line has `per_rank` of length N, says how the key was replicated and at what rate, and calls a
    sharded timing MEASURED only when every shard sat on a physical GPU of its own (distinct ordinals and distinct PCI bus ids)."""
    b = _bench()
    d = json.load(open(os.path.join(ROOT, "profiles", "r5_08_bench_ranks8_one_gpu.json")))     # eight ranks rehearsed on the one GPU
    assert d["n_gpus"] == 8 and len(d["per_rank"]) == 8 and sorted(p["rank"] for p in d["per_rank"]) == list(range(8))
    assert sum(p["proofs"] for p in d["per_rank"]) == d["steps"] * 8 and d["scaling"] == "weak"
    cfg = _driver_view(d["config"])
    assert cfg["key_replication"] in ("nccl", "rccl", "gloo", "per-rank") and isinstance(cfg["key_bcast_GBps"], float) and cfg["key_bcast_GBps"] > 0
    assert cfg["sharded_parts"] == 8 and cfg["sharded_measured"] is False and "one device" in cfg["sharded_reason"]
    one = "0000:f4:00.0"
    assert b.shards_share_a_gpu([0, 0], [one, one]) is True                       # the rehearsal: one device twice
    assert b.shards_share_a_gpu([0, 1], [one, one]) is True                       # two ordinals that are ONE physical GPU (same bus id)
    assert b.shards_share_a_gpu([0, 1], [one, ""]) is True                        # a bus id that could not be read decides nothing
    assert b.shards_share_a_gpu([0, 1, 2, 3], ["0000:%02x:00.0" % (0x10 + i) for i in range(4)]) is False
    leg = dict(d["intra_proof_sharding"], rehearsal_on_one_gpu=b.shards_share_a_gpu([0, 1], [one, "0000:f5:00.0"]))
    assert b.sharding_scalars(leg)["sharded_measured"] is True


def test_round_6_default_line_carries_ratios_counters_and_denominators():
    """The default line as recorded on the round's last tree (profiles/r6_21_bench_default.json): the scalars VERDICT r5 asked for are
    there, consistent with each other, and the counter files they come from are THIS round's."""
    b = _bench()
    d = json.load(open(os.path.join(ROOT, "profiles", "r6_21_bench_default.json")))
    roof, cpu, cfg = _driver_view(d["roofline"]), _driver_view(d["cpu_baseline"]), _driver_view(d["config"])
    for obj in (d["config"], d["roofline"], d["cpu_baseline"]):
        flat = [k for k, v in obj.items() if not isinstance(v, (dict, list))]
        assert len({k[:40] for k in flat}) == len(flat)
    # roofline: the contract form, the calibrated traffic and its ratios, the counter-based busy fraction
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and roof["kernel"] == "msm_accum_kernel<Fq>" and roof["launches_per_proof"] == 4.0
    assert abs(roof["traffic_over_algorithmic"] - roof["traffic"] / roof["algorithmic_bytes_per_launch"]) < 1e-9 and 5 < roof["traffic_over_algorithmic"] < 15
    assert abs(roof["hbm_traffic_over_algorithmic_proof"] - roof["hbm_traffic_GB_per_proof"] / roof["algorithmic_GB_per_proof"]) < 1e-9
    assert "r%d_pmc_traffic.json" % b.ROUND in roof["traffic_source"] and "x 0.797" in roof["traffic_source"]      # inside the 120 characters the driver keeps
    busy = 4.0 * roof["valu_wave_instructions_per_proof"] / (1024 * roof["sclk_mhz_sampled"] * 1e6 * d["ms_per_step"] * 1e-3)
    assert abs(roof["valu_busy_step"] - busy) < 1e-9 and 0.7 < busy < 0.9
    # the committed counter files bench.py reads are this round's, and agree with what the line says
    pmc = json.load(open(os.path.join(ROOT, "profiles", b.PMC_FILE)))
    cen = json.load(open(os.path.join(ROOT, "profiles", b.CENSUS_FILE)))
    assert pmc["round"] == b.ROUND == cen["round"] and cen["valu_wave_instructions_per_proof"] == roof["valu_wave_instructions_per_proof"]
    assert pmc["kernels"]["msm_accum_kernel<Fq>"]["pattern"] == "gather64" and pmc["kernels"]["msm_accum_kernel<Fq2>"]["pattern"] == "gather128"
    assert abs(pmc["kernels"]["msm_accum_kernel<Fq>"]["hbm_bytes_per_launch"] - roof["traffic"]) < 1.0
    # cpu_baseline: the north star's denominator and ratios
    assert cpu["snarkjs_style_sample_log_m"] == 12 and cpu["speedup_vs_snarkjs_style"] > 50 and cpu["cpu_and_gpu_proofs_identical"] is True
    assert abs(cpu["speedup_vs_c_1thread"] * cpu["value"] - d["value"]) < 1e-9 and cpu["all_threads_cores"] >= 1
    # config: the reference's calling pattern and the other BASELINE config
    for k in ("sync_latency_ms", "host_buffer_sync_proofs_per_s", "rate_2_22_proofs_per_s", "tx_single_proof_ms", "tx_fused_proofs_per_s", "withdraw_single_proof_ms", "sharded_ms"):
        assert isinstance(cfg.get(k), float), k
    assert d["metric"].startswith("Groth16 proofs/sec") and d["unit"] == "proofs/s" and d["dtype"] and d["vs_baseline"] is None and d["scaling"] == "weak"
