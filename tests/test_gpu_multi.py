"""Several GPUs of one node from ONE process through the C ABI (SURVEY.md 8(b) "Threading", 8(e); BASELINE configs[3]):
zkr_key_replicate + zkr_prove_batch_multi, and the hosts on top of them (facade.Bn128.groth16GenProofBatch(devices=...),
index.js groth16GenProofBatch(..., {devices})).  The GPU box has one device, so the replicas live side by side on device 0
(devices = [0, 0]): replication (both forms), sharding, the per-key host threads and the gather are all exercised; with
more devices visible the same tests also spread over them."""
import ctypes
import json

import pytest

import groth16 as g

pytestmark = pytest.mark.gpu


def _arena_bytes(key):
    import torch
    from zkr_hip.batch import _tensor_from_ptr
    ptr, n = key.arena()
    return _tensor_from_ptr(ptr, n, key.device)


def _devices(n):
    import zkr_hip
    have = zkr_hip.device_count()
    return [i % have for i in range(n)]


@pytest.mark.parametrize("log_m,count,mode", [(13, 16, "full"), (13, 5, "base"), (17, 16, "auto"), (20, 6, "full")])
def test_batch_over_replicas_equals_closed_form_and_single_key_batch(log_m, count, mode):
    """16 proofs over two replicas == the toxic-waste closed form == the same batch on the one source key; the replica's
    arena is the source's bytes in every mode."""
    import torch
    import zkr_hip
    p = 73
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    devs = _devices(2)
    rep = key.replicate(devs[1], mode)
    assert rep.device == devs[1] and rep.info() == key.info() and rep.windows() == key.windows()
    assert zkr_hip.lib().zkr_key_device(rep._h) == devs[1]
    if log_m <= 17:
        assert torch.equal(_arena_bytes(key).cpu(), _arena_bytes(rep).cpu())
    wbs = [wb] + [zkr_hip.synth_witness(log_m, p, 0x5A4B0001, 9100 + i) for i in range(1, 4)]
    wbs = [wbs[i % len(wbs)] for i in range(count)]
    rng = g.SplitMix64(1000 + log_m + count)
    rs, ss = [rng.fr() for _ in range(count)], [rng.fr() for _ in range(count)]
    multi = zkr_hip.prove_batch_multi([key, rep], wbs, rs, ss)
    single = key.prove_batch(wbs, rs, ss)
    assert multi == single
    for i in ([0, 1, 2, count - 1] if log_m >= 17 else range(count)):
        expect, _, _ = g.proof_from_aux(aux, wbs[i], p, rs[i], ss[i])
        assert multi[i] == g.proof_bytes(expect), "proof %d" % i
    # resident witnesses: witness i on the device of keys[i mod 2]
    keys = [key, rep]
    dw = [torch.frombuffer(bytearray(wbs[i]), dtype=torch.uint8).to(torch.device("cuda", keys[i % 2].device)) for i in range(count)]
    for d in set(devs):
        torch.cuda.synchronize(d)
    assert zkr_hip.prove_batch_multi_device(keys, [t.data_ptr() for t in dw], rs, ss) == single
    # random blinding: every proof verifies, no two alike; the source key may go first -- a replica is independent of it
    vk_bin = key.synth_vk(aux)
    key.close()
    rep2 = rep.replicate(devs[0], "full")
    rnd = zkr_hip.prove_batch_multi([rep, rep2], wbs[:7])
    pubs = [[int.from_bytes(w[32 * j:32 * j + 32], "little") for j in range(1, p + 1)] for w in wbs[:7]]
    assert zkr_hip.verify_batch(vk_bin, rnd, pubs) and len(set(rnd)) == len(rnd)


def test_three_replicas_ragged_counts_and_errors():
    import zkr_hip
    log_m, p = 10, 73
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    devs = _devices(3)
    keys = [key, key.replicate(devs[1], "base"), key.replicate(devs[2], "full")]
    rng = g.SplitMix64(31337)
    for count in (1, 2, 3, 4, 11):
        rs, ss = [rng.fr() for _ in range(count)], [rng.fr() for _ in range(count)]
        out = zkr_hip.prove_batch_multi(keys, [wb] * count, rs, ss)
        assert out == [key.prove(wb, r, s) for r, s in zip(rs, ss)]
    assert zkr_hip.prove_batch_multi(keys, []) == []
    with pytest.raises(zkr_hip.ZkrError, match="listed twice"):
        zkr_hip.prove_batch_multi([key, key], [wb, wb])
    other, wb2, _ = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FE)          # same circuit, another setup
    with pytest.raises(zkr_hip.ZkrError, match="not a replica"):
        zkr_hip.prove_batch_multi([key, other], [wb, wb])
    with pytest.raises(zkr_hip.ZkrError) as e:
        zkr_hip.prove_batch_multi(keys, [wb[:-32]] * 3)                                  # every share refuses the length
    assert e.value.code == -3
    with pytest.raises(zkr_hip.ZkrError, match=r"key \d+ \(device \d+\)"):
        zkr_hip.prove_batch_multi(keys, [wb] * 4, [1, 2, g.R, 4], [1, 2, 3, 4])          # one share fails: its message names the key
    with pytest.raises(zkr_hip.ZkrError):
        key.replicate(zkr_hip.device_count() + 3)
    h = ctypes.c_void_p()
    assert zkr_hip.lib().zkr_key_replicate(key._h, 0, 7, ctypes.byref(h)) == -5


def test_concurrent_multi_batches_and_sharded_proofs_on_the_same_keys():
    """Two host threads inside zkr_prove_batch_multi on the SAME replicas at once (two libuv workers behind two
    groth16GenProofBatch calls), a third proving sharded on shards of the same key: the per-key slot hand-out serialises
    nothing wrongly and every proof is the closed form's."""
    import threading
    import zkr_hip
    log_m, p = 13, 73
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    devs = _devices(2)
    rep = key.replicate(devs[1])
    shards = [key.shard(i, 2, devs[i]) for i in range(2)]
    wbs = [wb] + [zkr_hip.synth_witness(log_m, p, 0x5A4B0001, 8800 + i) for i in range(1, 3)]
    out, errs = {}, []

    def batch(tag, base):
        try:
            n = 10
            ws = [wbs[(base + i) % 3] for i in range(n)]
            rs, ss = [base + 7 * i + 1 for i in range(n)], [base + 11 * i + 3 for i in range(n)]
            out[tag] = (ws, rs, ss, zkr_hip.prove_batch_multi([key, rep], ws, rs, ss))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    def sharded(tag):
        try:
            out[tag] = [zkr_hip.prove_sharded(shards, wbs[i % 3], 500 + i, 900 + i) for i in range(6)]
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=batch, args=("a", 100)), threading.Thread(target=batch, args=("b", 2000)), threading.Thread(target=sharded, args=("s",))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for tag in ("a", "b"):
        ws, rs, ss, proofs = out[tag]
        for w, r, s_, got in zip(ws, rs, ss, proofs):
            assert got == g.proof_bytes(g.proof_from_aux(aux, w, p, r, s_)[0])
    for i, got in enumerate(out["s"]):
        assert got == g.proof_bytes(g.proof_from_aux(aux, wbs[i % 3], p, 500 + i, 900 + i)[0])


def test_facade_batch_over_devices_parses_the_key_once(small_case):
    """facade.Bn128.groth16GenProofBatch(devices=[0, 0]) from the websnark buffer, as the reference's caller holds it
    (common.ts:28): one parse + one device-to-device replication, afterwards only cache hits; proofs == oracle."""
    import coracle
    import zkr_hip
    c = small_case
    zkr_hip.clear_key_cache()
    before = dict(zkr_hip.key_cache_stats)
    rng = g.SplitMix64(4)
    n = 9
    rs, ss = [rng.fr() for _ in range(n)], [rng.fr() for _ in range(n)]
    devs = _devices(2)
    expect = [zkr_hip.proof_json_from_bytes(coracle.prove(c["pkb"], c["wb"], r, s)) for r, s in zip(rs, ss)]
    for _ in range(2):
        bn = zkr_hip.build_bn128()
        assert bn.groth16GenProofBatch([c["wb"]] * n, bytes(c["pkb"]), rs, ss, devices=devs) == expect
    d = {k: zkr_hip.key_cache_stats[k] - before[k] for k in ("loads", "replications", "hits")}
    assert d == {"loads": 1, "replications": 1, "hits": 2}
    assert zkr_hip.build_bn128().groth16GenProofBatch([c["wb"]] * 3, bytes(c["pkb"]), rs[:3], ss[:3]) == expect[:3]   # single device: the first replica
    # ONE proof over the devices: shards cut from the cached key, cached themselves
    for i in range(2):
        assert zkr_hip.build_bn128().groth16GenProof(c["wb"], bytes(c["pkb"]), rs[i], ss[i], devices=_devices(3)) == expect[i]
    assert zkr_hip.key_cache_stats["loads"] - before["loads"] == 1
    zkr_hip.clear_key_cache()


def _scaled(circ, rows_to_scale):
    """The same circuit with rows (2A) B = (2C): same witness, same sizes and term counts, other coefficients."""
    rows = list(circ["rows"])
    for i in rows_to_scale:
        A, B, C = rows[i]
        rows[i] = ([(s, cf * 2 % g.R) for s, cf in A], B, [(s, cf * 2 % g.R) for s, cf in C])
    out = dict(circ)
    out["rows"] = rows
    return out


def _r1cs(circ):
    import zkr_hip
    return zkr_hip.binarify_r1cs(dict(nVars=circ["nVars"], nPublic=circ["nPublic"], constraints=[[dict(lc) for lc in row] for row in circ["rows"]]))


def test_key_cache_does_not_alias_equal_sized_circuits_of_one_setup():
    """two circuits of equal (n, p, m, nnz) set up from the SAME toxic waste share the header points
    and the hExps tail of the provingKeyBin; the cache must still tell them apart (the reference re-parses on every call,
    common.ts:28).  Sampled fingerprint: a circuit that differs broadly; ZKR_KEY_FINGERPRINT=full semantics
    (key_fingerprint(buf, full=True)): a single edited constraint."""
    import zkr_hip
    m, p = 1 << 11, 7
    circ = g.synth_circuit(m, p, 0x5A4B0021)
    broad = _scaled(circ, range(900, circ["nConstraints"], 3))   # from 900 on: the first 4 KiB (header + the first terms of signal 0) stay equal
    narrow = _scaled(circ, [circ["nConstraints"] // 2 + 1])
    assert g.check_r1cs(broad) and g.check_r1cs(narrow)
    tox = g.toxic_from_seed(0x5A4B00FF)
    toxl = [tox[k] for k in ("t", "alfa", "beta", "gamma", "delta")]
    pk0, _ = zkr_hip.setup_r1cs_websnark(_r1cs(circ), toxl)
    pk1, _ = zkr_hip.setup_r1cs_websnark(_r1cs(broad), toxl)
    pk2, _ = zkr_hip.setup_r1cs_websnark(_r1cs(narrow), toxl)
    assert len(pk0) == len(pk1) == len(pk2) > 4096 * 66                    # the sampled path, not the whole-buffer one
    assert pk0[:4096] == pk1[:4096] == pk2[:4096] and pk0[-4096:] == pk1[-4096:] == pk2[-4096:]   # what round 3 hashed: aliases
    assert zkr_hip.key_fingerprint(pk0, full=False) != zkr_hip.key_fingerprint(pk1, full=False)
    assert len({zkr_hip.key_fingerprint(k, full=True) for k in (pk0, pk1, pk2)}) == 3
    # round 5: the cache finds candidates by the sampled digest and CONFIRMS them by comparing the buffers, so the lone edited
    # constraint (which the sampled digest may or may not see) gets its own key either way -- checked on the proofs below
    wb = g.binarify_witness(circ["witness"])
    rng = g.SplitMix64(8)
    r, s = rng.fr(), rng.fr()
    zkr_hip.clear_key_cache()
    before = dict(zkr_hip.key_cache_stats)
    got0 = zkr_hip.groth16_gen_proof(wb, pk0, r=r, s=s)
    got1 = zkr_hip.groth16_gen_proof(wb, pk1, r=r, s=s)
    assert zkr_hip.key_cache_stats["loads"] - before["loads"] == 2 and got0 != got1
    assert got0 == g.proof_to_json(g.proof_from_toxic(circ, tox, circ["witness"], r, s))
    assert got1 == g.proof_to_json(g.proof_from_toxic(broad, tox, circ["witness"], r, s))
    # the single edited constraint: a third key for the cache, its own proof (round 4 proved with pk0's key here unless ZKR_KEY_FINGERPRINT=full)
    got2 = zkr_hip.groth16_gen_proof(wb, pk2, r=r, s=s)
    assert zkr_hip.key_cache_stats["loads"] - before["loads"] == 3 and got2 not in (got0, got1)
    assert got2 == g.proof_to_json(g.proof_from_toxic(narrow, tox, circ["witness"], r, s))
    zkr_hip.clear_key_cache()
