"""BASELINE.json full-size configuration (m = 2^20) through size-independent properties: the proof must
equal the toxic-waste closed form bit-for-bit and satisfy the TxVerifier.sol pairing equation."""
import pytest

import groth16 as g

pytestmark = pytest.mark.gpu


def _proof_points(pb):
    v = [int.from_bytes(pb[32 * i:32 * i + 32], "little") for i in range(8)]
    return dict(pi_a=(v[0], v[1]), pi_b=((v[2], v[3]), (v[4], v[5])), pi_c=(v[6], v[7]))


@pytest.mark.parametrize("log_m", [16, 20, 22])  # BASELINE.json configs[1] (2^20) and configs[2] (2^22)
def test_fullsize_proof_equals_closed_form_and_verifies(log_m):
    import zkr_hip
    p = 73
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    rng = g.SplitMix64(99 + log_m)
    r, s = rng.fr(), rng.fr()
    proof = key.prove(wb, r, s)
    expect, vk, pub = g.proof_from_aux(aux, wb, p, r, s)
    assert proof == g.proof_bytes(expect)
    if log_m == 16:  # the closed form with the C dot products (used at 2^24, tests/test_gpu_configs.py) is the same function
        import coracle
        assert g.proof_from_aux(aux, wb, p, r, s, dot=coracle.fr_dot) == (expect, vk, pub)
    assert g.is_valid(vk, _proof_points(proof), pub)
    bad = list(pub)
    bad[3] = (bad[3] + 1) % g.R
    assert not g.is_valid(vk, _proof_points(proof), bad)  # withdrawverifier.test.ts:42-68 pattern
    # the product's native verifier (zkr_verify) on the vk the product exports: same verdicts as the oracle's pairing
    vk_bin = key.synth_vk(aux)
    assert vk_bin == zkr_hip.binarify_verifying_key(vk)
    assert zkr_hip.verify(vk_bin, proof, pub) is True and zkr_hip.verify(vk_bin, proof, bad) is False
    # a second witness of the same circuit on the same key
    wb2 = zkr_hip.synth_witness(log_m, p, 0x5A4B0001, 4242)
    proof2 = key.prove(wb2, r, s)
    expect2, _, pub2 = g.proof_from_aux(aux, wb2, p, r, s)
    assert proof2 == g.proof_bytes(expect2) and proof2 != proof
    assert g.is_valid(vk, _proof_points(proof2), pub2)


@pytest.mark.parametrize("log_m", [12, 18])
def test_pipelined_batch_equals_synchronous_proofs(log_m):
    """zkr_prove_submit / zkr_prove_collect with two proofs in flight: every proof of the batch is bit-identical
    to the synchronous call with the same witness and blinding (and so to the closed form)."""
    import torch
    import zkr_hip
    p = 73
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    wbs = [wb] + [zkr_hip.synth_witness(log_m, p, 0x5A4B0001, 7000 + i) for i in range(4)]
    rng = g.SplitMix64(5 + log_m)
    rs, ss = [rng.fr() for _ in wbs], [rng.fr() for _ in wbs]
    dw = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wbs]
    batch = key.prove_batch_device([t.data_ptr() for t in dw], rs, ss)
    assert len(batch) == len(wbs)
    for w, r, s, got in zip(wbs, rs, ss, batch):
        assert got == key.prove(w, r, s)
        expect, _, _ = g.proof_from_aux(aux, w, p, r, s)
        assert got == g.proof_bytes(expect)
    # a third submit with two in flight is refused; collecting frees the slot
    t0 = key.prove_submit(dw[0].data_ptr(), rs[0], ss[0])
    t1 = key.prove_submit(dw[1].data_ptr(), rs[1], ss[1])
    with pytest.raises(zkr_hip.ZkrError):
        key.prove_submit(dw[2].data_ptr(), rs[2], ss[2])
    assert key.prove_collect(t0) == batch[0]
    t2 = key.prove_submit(dw[2].data_ptr(), rs[2], ss[2])
    assert key.prove_collect(t1) == batch[1]
    assert key.prove_collect(t2) == batch[2]
    with pytest.raises(zkr_hip.ZkrError):
        key.prove_collect(t2)


@pytest.mark.parametrize("log_m,count", [(9, 1), (9, 2), (10, 7), (13, 16), (13, 37), (16, 20), (18, 9)])
def test_fused_small_circuit_batches_equal_single_proofs(log_m, count):
    """zkr_prove_batch_device / zkr_prove_batch on circuits far below the chip's size (the reference's tx circuit is 2^17):
    up to fuse() proofs share every launch -- witnesses end to end, one bucket set per proof and table, the bucket
    reductions of the group in one launch -- and every proof must still be the bytes of the one-at-a-time path and of the
    toxic-waste closed form.  Counts cover a single proof, a batch cut in two halves, full groups plus a ragged tail."""
    import torch
    import zkr_hip
    p = 73 if log_m >= 10 else 5
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    cap = key.fuse()
    assert cap == max(1, min(16, (1 << 20) >> log_m, 256 // max(1, (1 << (min(log_m, 20) - 1)) // 8192)))
    wbs = [wb] + [zkr_hip.synth_witness(log_m, p, 0x5A4B0001, 31000 + i) for i in range(1, min(count, 5))]
    wbs = [wbs[i % len(wbs)] for i in range(count)]
    rng = g.SplitMix64(17 * log_m + count)
    rs, ss = [rng.fr() for _ in range(count)], [rng.fr() for _ in range(count)]
    dw = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wbs[:5]]
    dev = key.prove_batch_device([dw[i % len(dw)].data_ptr() for i in range(count)], rs, ss)
    host = key.prove_batch(wbs, rs, ss)
    assert len(dev) == len(host) == count
    closed = {}
    for i in range(count):
        assert dev[i] == host[i], "proof %d: device-witness batch and host-witness batch differ" % i
        if i < 6 or i >= count - 2:                                       # the one-at-a-time path and the closed form on a sample
            assert dev[i] == key.prove(wbs[i], rs[i], ss[i])
            expect, _, _ = g.proof_from_aux(aux, wbs[i], p, rs[i], ss[i])
            assert dev[i] == g.proof_bytes(expect)
    # random blinding inside a fused batch: every proof verifies and no two are alike
    vk_bin = key.synth_vk(aux)
    rnd = key.prove_batch(wbs[:min(count, cap + 1)])
    pubs = [[int.from_bytes(w[32 * j:32 * j + 32], "little") for j in range(1, p + 1)] for w in wbs[:len(rnd)]]
    assert zkr_hip.verify_batch(vk_bin, rnd, pubs) and len(set(rnd)) == len(rnd)


def test_concurrent_host_buffer_callers_on_one_key():
    """Several host threads inside zkr_prove on ONE key at once (libuv workers behind concurrent groth16GenProof calls):
    every caller uploads its pageable witness into its own staging buffer outside the key's lock, at most two proofs
    compute, the others wait for a slot -- and every proof is the bytes of the closed form for its witness and blinding."""
    import threading
    import zkr_hip
    log_m, p = 14, 73
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    wbs = [wb] + [zkr_hip.synth_witness(log_m, p, 0x5A4B0001, 5100 + i) for i in range(1, 4)]
    n_thr, per = 5, 6
    out, errs = {}, []

    def worker(j):
        try:
            for i in range(per):
                w = wbs[(j + i) % len(wbs)]
                out[(j, i)] = key.prove(w, 1000 + 10 * j + i, 2000 + 10 * j + i)
        except Exception as e:       # surfaced below: an exception in a thread must fail the test
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(j,)) for j in range(n_thr)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert len(out) == n_thr * per
    for (j, i), proof in out.items():
        expect, _, _ = g.proof_from_aux(aux, wbs[(j + i) % len(wbs)], p, 1000 + 10 * j + i, 2000 + 10 * j + i)
        assert proof == g.proof_bytes(expect), (j, i)


@pytest.mark.timeout(600)
def test_two_keys_prove_at_once_on_the_devices_shared_streams():
    """The streams belong to the device, not to the key (zkr_key.hip DeviceStreams): two keys of different sizes proving at the same
    time from different threads -- single calls and batch calls, pipelined two deep each -- enqueue onto the same five streams.
    The device's enqueue lock keeps one proof's launches together, so the cross-stream waits cannot form a circle (a hang fails by
    timeout), and every proof is the closed form's bytes for its key, witness and blinding; a key freed in between (its free
    synchronises the shared streams) disturbs nothing."""
    import threading
    import zkr_hip
    p = 73
    keys = {}
    for log_m in (12, 14):
        key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
        keys[log_m] = (key, [wb, zkr_hip.synth_witness(log_m, p, 0x5A4B0001, 7100 + log_m)], aux)
    out, errs = {}, []

    def worker(log_m, j, batch):
        key, wbs, _ = keys[log_m]
        try:
            for i in range(5):
                if batch:
                    ws = [wbs[(i + q) % 2] for q in range(5)]
                    rs, ss = [3000 + 100 * j + 10 * i + q for q in range(5)], [4000 + 100 * j + 10 * i + q for q in range(5)]
                    for q, pr in enumerate(key.prove_batch(ws, rs, ss)):
                        out[(log_m, j, i, q)] = (pr, (i + q) % 2, rs[q], ss[q])
                else:
                    out[(log_m, j, i, 0)] = (key.prove(wbs[i % 2], 3000 + 100 * j + i, 4000 + 100 * j + i), i % 2, 3000 + 100 * j + i, 4000 + 100 * j + i)
        except Exception as e:
            errs.append(e)

    def churn():  # a third key comes and goes while the others prove
        try:
            for _ in range(3):
                k3, w3, _a = zkr_hip.ProvingKey.synth(10, p, 0x5A4B0001, 0x5A4B00FF)
                k3.prove(w3, 5, 7)
                k3.close()
        except Exception as e:
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(12, 0, False)), threading.Thread(target=worker, args=(14, 1, False)),
           threading.Thread(target=worker, args=(12, 2, True)), threading.Thread(target=worker, args=(14, 3, True)), threading.Thread(target=churn)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert len(out) == 2 * 5 + 2 * 25
    for (log_m, j, i, q), (proof, wi, r, s_) in out.items():
        _, wbs, aux = keys[log_m]
        expect, _, _ = g.proof_from_aux(aux, wbs[wi], p, r, s_)
        assert proof == g.proof_bytes(expect), (log_m, j, i, q)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mix", ["batch_vs_single", "batch_vs_batch"])
def test_batch_calls_do_not_deadlock_against_other_host_buffer_callers(mix):
    """zkr_prove_batch used to wait for a third staging buffer while its own groups held both proof slots
    and two staging buffers; a concurrent zkr_prove (or a second batch) holding the last buffer and waiting for a slot then
    hung both for good.  A batch call now collects its own oldest group instead of waiting.  Batches of >= 4 groups against
    single callers, and batch against batch, on ONE key from several threads; a hang fails by timeout, and every proof must
    still be the closed form's bytes."""
    import threading
    import zkr_hip
    log_m, p = 14, 73
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    fuse = max(1, key.fuse())
    wbs = [wb] + [zkr_hip.synth_witness(log_m, p, 0x5A4B0001, 6100 + i) for i in range(1, 3)]
    n_batch = 6 * max(fuse, 1) + 1          # >= 4 groups whatever the key's fused capacity, ragged last group
    out, errs = {}, []

    def batch_worker(j, rounds):
        try:
            for q in range(rounds):
                ws = [wbs[(j + q + i) % len(wbs)] for i in range(n_batch)]
                rs = [3000 + 1000 * j + 100 * q + i for i in range(n_batch)]
                ss = [7000 + 1000 * j + 100 * q + i for i in range(n_batch)]
                for i, pr in enumerate(key.prove_batch(ws, rs, ss)):
                    out[("b", j, q, i)] = (pr, (j + q + i) % len(wbs), rs[i], ss[i])
        except Exception as e:
            errs.append(e)

    def single_worker(j, count):
        try:
            for i in range(count):
                r, sc = 500 + 50 * j + i, 900 + 50 * j + i
                out[("s", j, i)] = (key.prove(wbs[(j + i) % len(wbs)], r, sc), (j + i) % len(wbs), r, sc)
        except Exception as e:
            errs.append(e)

    if mix == "batch_vs_single":
        ths = [threading.Thread(target=batch_worker, args=(0, 3))] + [threading.Thread(target=single_worker, args=(j, 12)) for j in range(1, 4)]
    else:
        ths = [threading.Thread(target=batch_worker, args=(j, 3)) for j in range(3)] + [threading.Thread(target=single_worker, args=(3, 8))]
    for t in ths:
        t.daemon = True
        t.start()
    for t in ths:
        t.join(300)
        assert not t.is_alive(), "a caller is stuck: hold-and-wait between proof slots and staging buffers"
    assert not errs, errs
    sample = sorted(out)[::7]
    for k in sample:
        proof, wi, r, sc = out[k]
        expect, _, _ = g.proof_from_aux(aux, wbs[wi], p, r, sc)
        assert proof == g.proof_bytes(expect), k


def test_off_curve_points_are_refused_when_point_checks_are_on(monkeypatch):
    """ZKR_CHECK_POINTS=1 runs an on-curve kernel over every table's base points at key load.  A
    provingKeyBin with ONE coordinate of one G1 point (pointsA) or of one G2 point (pointsB2) edited is refused with the
    table and the point named; the clean buffer loads with the check on; with the check off (the default, and what the
    reference does: websnark multiplies whatever the key holds) the edited buffer still loads."""
    import struct
    import zkr_hip
    pkb, wb = zkr_hip.synth_websnark(9, 5, 0x5A4B0001, 0x5A4B00FF)
    ptr_a, ptr_b2 = struct.unpack_from("<I", pkb, 20)[0], struct.unpack_from("<I", pkb, 28)[0]

    def edited(ptr, size):
        b = bytearray(pkb)
        j = next(j for j in range(2, 64) if any(b[ptr + size * j:ptr + size * j + size // 2]))   # a finite point (x != 0)
        b[ptr + size * j + size // 2] ^= 1                                                      # lowest bit of y (y.re for G2)
        return bytes(b), j

    monkeypatch.setenv("ZKR_CHECK_POINTS", "1")
    key = zkr_hip.ProvingKey.load_websnark(pkb)
    want = key.prove(wb, 5, 7)
    key.close()
    for ptr, size, group in ((ptr_a, 64, "G1"), (ptr_b2, 128, "G2")):
        bad, j = edited(ptr, size)
        with pytest.raises(zkr_hip.ZkrError, match="%s table are not on the curve" % group):
            zkr_hip.ProvingKey.load_websnark(bad)
    monkeypatch.delenv("ZKR_CHECK_POINTS")
    bad, _ = edited(ptr_a, 64)
    k2 = zkr_hip.ProvingKey.load_websnark(bad)          # unchecked, as in the reference: loads, and proves something else
    assert k2.prove(wb, 5, 7) != want
    k2.close()


def test_websnark_buffer_path_at_2_16():
    """The reference's own data flow at the size of the real tx circuit's neighbourhood (2^17): a
    60 MB provingKeyBin in the binarify.ts layout through zkr_key_load_websnark, proof == the C oracle on the same
    buffers, accepted by the native verifier of the matching setup."""
    import coracle
    import zkr_hip
    log_m, p = 16, 73
    pkb, wb = zkr_hip.synth_websnark(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    key = zkr_hip.ProvingKey.load_websnark(pkb)
    info = key.info()
    assert info["domainSize"] == 1 << log_m and info["nPublic"] == p
    rng = g.SplitMix64(1616)
    r, s = rng.fr(), rng.fr()
    proof = key.prove(wb, r, s)
    assert proof == coracle.prove(pkb, wb, r, s)
    key2, wb2, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    assert wb2 == wb and key2.prove(wb, r, s) == proof
    pub = [int.from_bytes(wb[32 * i:32 * i + 32], "little") for i in range(1, p + 1)]
    assert zkr_hip.verify(key2.synth_vk(aux), proof, pub)


def test_fullsize_arena_replica_above_4_gib():
    """The multi-GPU path's hand-off at its real size: the 4.5 GB arena (window tables included) viewed as a torch
    tensor, copied (stands in for the RCCL broadcast) and adopted as a second key proves identically."""
    import torch
    import zkr_hip
    from zkr_hip.batch import _tensor_from_ptr
    key, wb, _ = zkr_hip.ProvingKey.synth(20, 73, 0x5A4B0001, 0x5A4B00FF, want_aux=False)
    ptr, n = key.arena()
    assert n > (1 << 32)
    view = _tensor_from_ptr(ptr, n, 0)
    replica = view.clone()
    torch.cuda.synchronize()
    assert replica.numel() == n and bool((replica[-4096:] == view[-4096:]).all().item())
    key2 = zkr_hip.ProvingKey.adopt_arena(replica.data_ptr(), n, 0, keepalive=replica)
    assert key2.info() == key.info() and key2.windows() == key.windows()
    assert key2.prove(wb, 5, 7) == key.prove(wb, 5, 7)


@pytest.mark.parametrize("launcher", ["plain", "torchrun"])
def test_bench_two_rank_path_rehearsal_on_one_gpu(launcher):
    """The N > 1 flow of bench.py end to end on this one-GPU box: two ranks on cuda:0, gloo instead of RCCL (which
    refuses two ranks per device): key built on rank 0, arena broadcast, adopted by rank 1, both shards proved,
    max-over-ranks timing, one JSON line from rank 0 with the whole-job rate."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZKR_BENCH_BACKEND="gloo", ZKR_BENCH_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    # "plain": `python bench.py --gpus 2` starts its own ranks as a child torchrun;
    # "torchrun": the driver's launch line for N > 1
    cmd = [sys.executable]
    if launcher == "torchrun":
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    r = subprocess.run(cmd + [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--log-m", "14"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]      # whole-job rate: all ranks' proofs / max time
    assert d["key"]["bcast_s"] is not None and d["key"]["bcast_GBps"] > 0 and d["proofs_verified"] >= 1
    assert [(p["rank"], p["proofs"]) for p in d["per_rank"]] == [(0, 6), (1, 6)]
    # round 5: with N > 1 the intra-proof sharding leg is MEASURED -- rank 0 builds shard i on device i (here both on cuda:0)
    # and times zkr_prove_sharded_device; which form ran is in the line, as scalars of `config` too
    sh = d["intra_proof_sharding"]
    assert sh["measured"] is True and sh["parts"] == 2 and sh["devices"] == [0, 0] and sh["rehearsal_on_one_gpu"] is True
    assert sh["proof_identical_to_whole_key"] is True and sh["form"] == "split" and sh["ms_per_proof"] > 0 and sh["replicated_calch_ms_per_proof"] > 0
    assert len(sh["split_phase_ms_per_shard"]) == 2 and all(x > 0 for x in sh["remote_GBps_per_shard_in_cross_phases"])
    cfg = d["config"]
    assert cfg["sharded_parts"] == 2 and cfg["sharded_form"] == "split" and cfg["sharded_measured"] is False and cfg["sharded_ms"] == sh["ms_per_proof"]
    assert cfg["key_replication"] == "gloo" and cfg["key_bcast_GBps"] > 0


def test_bench_two_rank_rehearsal_survives_a_failed_broadcast():
    """the key broadcast fails on every rank (ZKR_FORCE_BCAST_FAIL=1), bench.py's ranks agree on it over
    the control-plane group, each builds its own replica and the line says so (`key.replication: "per-rank"`)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZKR_BENCH_BACKEND="gloo", ZKR_BENCH_ONE_GPU="1", ZKR_FORCE_BCAST_FAIL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--log-m", "13"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["key"]["replication"] == "per-rank" and d["key"]["bcast_GBps"] is None
    assert [(p["rank"], p["proofs"]) for p in d["per_rank"]] == [(0, 4), (1, 4)] and d["proofs_verified"] == 4
    assert "every rank builds its own replica" in r.stderr


@pytest.mark.parametrize("mode", ["auto", "base"])
def test_bench_inproc_multi_device_rehearsal_on_one_gpu(mode):
    """`python bench.py --gpus 2 --inproc`: N GPUs from ONE process through zkr_key_replicate +
    zkr_prove_batch_multi_device, no torch.distributed; on this one-GPU box both replicas sit on device 0 (--devices 0,0).
    Same JSON shape as the ranks' line; every timed proof goes through the native verifier."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--inproc", "--devices", "0,0", "--replicate-mode", mode,
                        "--steps", "5", "--warmup", "1", "--log-m", "14"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and d["proofs_verified"] == 10
    assert abs(d["value"] - 10 / (d["ms_per_step"] * 5e-3)) < 1e-6 * d["value"]
    assert d["config"]["devices"] == [0, 0] and "REHEARSAL" in d["config"]["parallelism"]
    assert len(d["key"]["replicas"]) == 1 and d["key"]["replicas"][0]["GBps"] > 0
    assert d["roofline"]["frac"] > 0 and set(d) >= {"metric", "value", "unit", "roofline", "config", "dtype", "data", "vs_baseline"}
    # round 5: how the replica was made and the measured sharded proof over the same device list, in the line and as scalars
    assert d["key"]["replica_modes"] == [{"mode": "full" if mode == "auto" else "base", "peer_direct": True}]
    assert d["config"]["key_replication"].startswith("peer-copy: " + ("full" if mode == "auto" else "base"))
    sh = d["intra_proof_sharding"]
    assert sh["measured"] is True and sh["devices"] == [0, 0] and sh["proof_identical_to_whole_key"] is True and sh["form"] == "split"
    assert d["config"]["sharded_parts"] == 2 and d["config"]["sharded_ms"] > 0 and d["config"]["sharded_reason"] == sh["reason"]
    assert d["roofline"]["valu_frac_kernel"] > 0 and d["roofline"]["valu_frac_proof"] > 0          # scalars the driver's record keeps


def test_multi_gpu_preflight_rehearsal_on_one_gpu():
    """tools/multi_gpu_preflight.py --devices 0,0: peer matrix, a device-to-device copy, the key replicated
    in both forms, a 2-shard sharded proof (first-use form reported, proof == whole key's, verifies), and the RCCL step -- which on
    ONE device must end in the per-rank fallback (RCCL refuses two ranks per device), printed, not hung."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "multi_gpu_preflight.py"), "--devices", "0,0", "--copy-mib", "256", "--log-m", "14", "--rccl-timeout", "20"],
                       capture_output=True, text=True, timeout=400, env=env, cwd=root)
    rows = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            d = json.loads(ln)
            rows[d["step"]] = d
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert set(rows) >= {"peers", "peer_copy", "replicate", "sharded", "rccl", "summary"}
    assert rows["peers"]["can_access"] == [[True]] and rows["peer_copy"]["pairs"][0]["intact"] and rows["peer_copy"]["pairs"][0]["GBps"] > 50
    assert [x["mode"] for x in rows["replicate"]["replicas"]] == ["full", "base"] and all(x["proof_identical"] for x in rows["replicate"]["replicas"])
    run = rows["sharded"]["runs"][0]
    assert run["parts"] == 2 and run["identical_to_whole_key"] and run["verifies"] and run["form"] == "split"
    assert rows["summary"]["ok"] is True and rows["summary"]["rehearsal"] is True
    assert "key_replication_would_use" in rows["summary"]     # rccl or the per-rank fallback: either way a diagnosis
