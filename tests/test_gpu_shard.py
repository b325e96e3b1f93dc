"""ONE proof over several GPUs (BASELINE configs[2] and [4] are single proofs): zkr_key_shard cuts
every MSM of the proof into contiguous ranges, every shard computes its partial sums, the host adds them and assembles.  On
the one-GPU box the shards are built side by side on device 0; the combined proof must be the bytes of
the whole key's proof and of the toxic-waste closed form."""
import pytest

import coracle
import groth16 as g

pytestmark = pytest.mark.gpu


def closed_form(aux, wb, p, r, s, log_m):
    """The toxic-waste closed form of the proof (no MSM, no NTT); above 2^18 the three dot products run in the C oracle
    (seconds instead of minutes of Python big integers), as tests/test_gpu_configs.py does at 2^24."""
    return g.proof_bytes(g.proof_from_aux(aux, wb, p, r, s, dot=coracle.fr_dot if log_m > 18 else None)[0])


@pytest.mark.parametrize("log_m,parts", [(7, 8), (10, 2), (13, 3), (16, 2), (16, 4), (16, 8), (20, 2), (20, 8)])   # (7, 8): five shards own no C point at all (73 public signals)
def test_sharded_proof_equals_whole_key_proof_and_closed_form(log_m, parts):
    import torch
    import zkr_hip
    p = 73
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    info = key.info()
    have = zkr_hip.device_count()
    shards = [key.shard(i, parts, device=i % have) for i in range(parts)]
    # the shards partition every scalar vector and every table
    cover_w = cover_h = 0
    pts = {t: 0 for t in ("ptsA", "ptsB1", "ptsB2", "ptsC", "ptsH")}
    for i, sh in enumerate(shards):
        si = sh.shard_info()
        assert (si["part"], si["parts"]) == (i, parts) and si["w_lo"] == cover_w and si["h_lo"] == cover_h
        cover_w += si["w_n"]
        cover_h += si["h_n"]
        for t in pts:
            pts[t] += sh.info()[t]
        # a shard picks its window from ITS scalar counts, as a key of that size would (csrc/zkr_key.hip msm_plan)
        c_of = lambda nsc: min(20, max(4, (nsc - 1).bit_length()))
        win = sh.windows()
        assert all(win[t][0] == c_of(si["w_n"]) for t in ("A", "B1", "B2", "C")) and win["H"][0] == c_of(si["h_n"]) and sh.fuse() == 1
        assert all(k == -(-255 // c) for c, k in win.values())
    assert cover_w == info["nVars"] and cover_h == info["domainSize"] and pts == {t: info[t] for t in pts}
    assert key.shard_info() == dict(part=0, parts=1, w_lo=0, w_n=info["nVars"], h_lo=0, h_n=info["domainSize"])
    rng = g.SplitMix64(77 + log_m + parts)
    r, s = rng.fr(), rng.fr()
    whole = key.prove(wb, r, s)
    expect, vk, pub = g.proof_from_aux(aux, wb, p, r, s)
    assert whole == g.proof_bytes(expect)
    # step by step: partial sums per shard, one after the other, then the combination (any shard or the whole key assembles)
    partials = [sh.prove_partial(wb) for sh in shards]
    assert all(len(x) == zkr_hip.PARTIAL_BYTES for x in partials)
    assert shards[-1].prove_combine(partials, r, s) == whole
    assert key.prove_combine(partials[::-1], r, s) == whole                  # a sum: the order of the records does not matter
    # in one call, shards concurrently (one host thread each); host witness and resident witnesses
    assert zkr_hip.prove_sharded(shards, wb, r, s) == whole
    dws = [torch.frombuffer(bytearray(wb), dtype=torch.uint8).to(torch.device("cuda", sh.device)) for sh in shards]
    torch.cuda.synchronize()
    assert zkr_hip.prove_sharded_device(shards, [t.data_ptr() for t in dws], r, s) == whole
    # another witness on the same shards; random blinding verifies
    wb2 = zkr_hip.synth_witness(log_m, p, 0x5A4B0001, 4242 + parts)
    assert zkr_hip.prove_sharded(shards, wb2, r, s) == key.prove(wb2, r, s)
    vk_bin = key.synth_vk(aux)
    rnd = zkr_hip.prove_sharded(shards, wb)
    assert rnd != whole and zkr_hip.verify(vk_bin, rnd, pub)
    # the whole key may go: shards own their memory
    key.close()
    assert zkr_hip.prove_sharded(shards, wb, r, s) == whole


@pytest.mark.parametrize("log_m,parts", [(12, 3), (12, 5), (12, 7), (18, 6), (22, 8)])
def test_ranges_of_h_that_are_not_aligned_blocks(log_m, parts):
    """A shard runs the last pair of calcH's transforms only on the aligned blocks that cover its range of h (csrc/zkr_prove.hip
    calc_h_device): ranges of m / 3, m / 5, ... start and end inside a block; 2^12 has two passes, 2^18 two, 2^22 three."""
    import zkr_hip
    p = 73
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    expect = closed_form(aux, wb, p, 21, 22, log_m)      # at EVERY size: the ranged transforms are held against the oracle directly,
    del aux                                              # not against the whole key's proof of the same library
    want = key.prove(wb, 21, 22)
    assert want == expect
    partials = []
    for i in range(parts):          # one shard at a time: a 2^22 key and eight shards side by side would not fit comfortably
        sh = key.shard(i, parts)
        partials.append(sh.prove_partial(wb))
        sh.close()
    assert key.prove_combine(partials, 21, 22) == expect


@pytest.mark.parametrize("log_m,parts", [(12, 2), (14, 4), (14, 8), (16, 8), (18, 4), (20, 8), (22, 8)])
def test_calc_h_split_over_the_shards(log_m, parts, monkeypatch):
    """zkr_prove_sharded with 2 / 4 / 8 shards splits calcH over them (csrc/zkr_prove.hip calc_h_split: every shard its block of the
    QAP rows and of every transform, the cross-block stages through the other shards' buffers): the same proof bytes as with h
    computed by every shard for itself, as the whole key's, as the closed form."""
    import torch
    import zkr_hip
    p = 73
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    want = closed_form(aux, wb, p, 31, 32, log_m)        # the oracle's bytes at every size, 2^20 and 2^22 included:
    del aux                                              # the CROSS passes are compared with them, not with this library's other route
    assert key.prove(wb, 31, 32) == want
    shards = [key.shard(i, parts) for i in range(parts)]
    assert zkr_hip.prove_sharded(shards, wb, 31, 32) == want
    stats = zkr_hip.sharded_split_stats()
    assert stats is not None and len(stats) == parts and all(ms > 0 for row in stats for ms in row[:4])
    form = zkr_hip.sharded_last_form()                   # WHICH form ran is reported, not inferred from the time
    assert form["form"] == "split" and "one device" in form["reason"]
    dw = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda(0)
    torch.cuda.synchronize()
    for _ in range(3):                                   # the barriers and the buffers survive being used again
        assert zkr_hip.prove_sharded_device(shards, [dw.data_ptr()] * parts, 31, 32) == want
    wb2 = zkr_hip.synth_witness(log_m, p, 0x5A4B0001, 777 + parts)
    assert zkr_hip.prove_sharded(shards, wb2, 5, 6) == key.prove(wb2, 5, 6)
    monkeypatch.setenv("ZKR_SHARD_SPLIT_H", "0")
    assert zkr_hip.prove_sharded(shards, wb, 31, 32) == want and zkr_hip.sharded_split_stats() is None
    assert zkr_hip.sharded_last_form() == {"form": "replicated", "reason": "replicated calcH: ZKR_SHARD_SPLIT_H=0"}
    monkeypatch.delenv("ZKR_SHARD_SPLIT_H")
    # three shards (not a power of two), or blocks too small for the cross passes: every shard for itself, silently
    if log_m == 12:
        three = [key.shard(i, 3) for i in range(3)]
        assert zkr_hip.prove_sharded(three, wb, 31, 32) == want and zkr_hip.sharded_split_stats() is None
        assert zkr_hip.sharded_last_form()["form"] == "replicated" and "3 shards" in zkr_hip.sharded_last_form()["reason"]
        eight = [key.shard(i, 8) for i in range(8)]      # 2^12 / 64 = 64 columns per shard: still split
        assert zkr_hip.prove_sharded(eight, wb, 31, 32) == want and zkr_hip.sharded_split_stats() is not None
    # two sharded proofs on the same shards at once take turns (their threads wait for one another inside the enqueue), a third
    # caller uses one of the shards on its own meanwhile
    if log_m == 16:
        import threading
        got, errs = {}, []
        def sharded(tag, r):
            try:
                got[tag] = [zkr_hip.prove_sharded(shards, wb, r, 32) for _ in range(4)]
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        def alone():
            try:
                got["alone"] = [shards[3].prove_partial(wb) for _ in range(6)]
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        ths = [threading.Thread(target=sharded, args=("x", 31)), threading.Thread(target=sharded, args=("y", 41)), threading.Thread(target=alone)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=120)
        assert not errs and not any(t.is_alive() for t in ths)
        assert got["x"] == [want] * 4 and got["y"] == [key.prove(wb, 41, 32)] * 4 and len(got["alone"]) == 6
    # a failing shard (short witness) takes the group down with an error, not with a hang
    with pytest.raises(zkr_hip.ZkrError):
        zkr_hip.prove_sharded(shards, wb[:-32], 31, 32)
    # ONE shard fails (no witness on its device) while the others are already waiting for it: the group is aborted, and the error
    # names the shard that failed, not one that gave up
    with pytest.raises(zkr_hip.ZkrError, match="shard 1 "):
        zkr_hip.prove_sharded_device(shards, [dw.data_ptr(), 0] + [dw.data_ptr()] * (parts - 2), 31, 32)
    assert zkr_hip.prove_sharded(shards, wb, 31, 32) == want


@pytest.mark.parametrize("log_m,parts", [(14, 8), (16, 4), (20, 8)])
def test_split_calc_h_with_a_witness_buffer_per_shard_between_poisoned_guards(log_m, parts):
    """Every shard of a split calcH gets its OWN copy of the witness (as on a node, where each lies on another device), placed
    at an odd 32-byte offset inside a larger allocation whose head and tail are poison (0xFF: values far above r, which the
    ingest would reduce into garbage): a QAP row, a cross pass or a digit kernel that reads one element outside its witness
    or mixes up the shards' buffers changes the proof.  Each shard's poison differs, and the run is repeated with the buffers
    permuted among the shards (same content, other addresses) and after the guards are rewritten."""
    import torch
    import zkr_hip
    p = 73
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    want = closed_form(aux, wb, p, 51, 52, log_m)
    del aux
    shards = [key.shard(i, parts) for i in range(parts)]
    n = len(wb)
    bufs, ptrs = [], []
    for i in range(parts):
        head, tail = 32 * (3 + 2 * i), 32 * (5 + i)
        t = torch.full((head + n + tail,), 0xFF - i, dtype=torch.uint8, device="cuda")
        t[head:head + n] = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda()
        bufs.append(t)
        ptrs.append(t.data_ptr() + head)
    torch.cuda.synchronize()
    assert zkr_hip.prove_sharded_device(shards, ptrs, 51, 52) == want
    assert zkr_hip.sharded_split_stats() is not None                      # it WAS the split form that ran
    assert zkr_hip.prove_sharded_device(shards, ptrs[1:] + ptrs[:1], 51, 52) == want
    for i, t in enumerate(bufs):                                          # other guard bytes, same witness
        head = 32 * (3 + 2 * i)
        t[:head] = 0x80 + i
        t[head + n:] = 0x7F - i
    torch.cuda.synchronize()
    assert zkr_hip.prove_sharded_device(shards, ptrs, 51, 52) == want
    # the positive control: ONE element of ONE shard's copy changed -- inside that shard's own range of w, so its MSM partial sums
    # move whatever the other shards computed -- must change the proof (the guards would be invisible otherwise)
    j = parts - 1
    si = shards[j].shard_info()
    at = 32 * (si["w_lo"] + si["w_n"] // 2)
    head = 32 * (3 + 2 * j)
    bufs[j][head + at] ^= 1
    torch.cuda.synchronize()
    assert zkr_hip.prove_sharded_device(shards, ptrs, 51, 52) != want


def test_first_use_check_of_the_split_form_and_its_fallback(monkeypatch, capfd):
    """Shards on DIFFERENT devices prove their first sharded proof both ways -- split calcH, then every shard for itself -- and keep
    the split only if the sums agree (the cross passes have never crossed a link).  On the one-GPU box the check is
    forced with ZKR_SHARD_SPLIT_CHECK=1; =2 also pretends the forms disagreed: the proof returned is the replicated form's (correct),
    a warning goes to stderr, and the shard set never splits again.  ZKR_SHARD_SPLIT_H=1 skips the check."""
    import zkr_hip
    log_m, p, parts = 14, 73, 4
    key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    want = closed_form(aux, wb, p, 61, 62, log_m)
    monkeypatch.setenv("ZKR_SHARD_SPLIT_CHECK", "1")
    good = [key.shard(i, parts) for i in range(parts)]
    assert zkr_hip.prove_sharded(good, wb, 61, 62) == want
    form = zkr_hip.sharded_last_form()
    assert form["form"] == "split" and "proved both ways" in form["reason"] and zkr_hip.sharded_split_stats() is not None
    assert zkr_hip.prove_sharded(good, wb, 61, 62) == want          # checked once: the second proof just splits
    assert zkr_hip.sharded_last_form()["form"] == "split" and "both ways" not in zkr_hip.sharded_last_form()["reason"]
    rnd = zkr_hip.prove_sharded(good, wb)                           # random blinding through the checked path
    assert zkr_hip.verify(key.synth_vk(aux), rnd, g.proof_from_aux(aux, wb, p, 1, 1)[2])
    # two callers meet a FRESH shard set at the same moment: both may run the first-use check (each proves both ways), both get the proof
    import threading
    fresh = [key.shard(i, parts) for i in range(parts)]
    got, forms = {}, {}
    def first_use(tag, r):
        got[tag] = zkr_hip.prove_sharded(fresh, wb, r, 62)
        forms[tag] = zkr_hip.sharded_last_form()                     # thread-local: each caller sees its own proof's form
    ths = [threading.Thread(target=first_use, args=(t, r)) for t, r in (("x", 61), ("y", 71))]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths)
    assert got["x"] == want and got["y"] == closed_form(aux, wb, p, 71, 62, log_m)
    assert forms["x"]["form"] == "split" and forms["y"]["form"] == "split" and any("both ways" in f["reason"] for f in forms.values())
    # the check belongs to the shard SET: a checked shard 0 with NEW siblings is an unchecked set and proves both ways again
    mixed = [good[0]] + [key.shard(i, parts) for i in range(1, parts)]
    assert zkr_hip.prove_sharded(mixed, wb, 61, 62) == want and "proved both ways" in zkr_hip.sharded_last_form()["reason"]
    assert zkr_hip.prove_sharded(mixed, wb, 61, 62) == want and "both ways" not in zkr_hip.sharded_last_form()["reason"]
    monkeypatch.setenv("ZKR_SHARD_SPLIT_CHECK", "2")
    bad = [key.shard(i, parts) for i in range(parts)]
    capfd.readouterr()
    assert zkr_hip.prove_sharded(bad, wb, 61, 62) == want           # the replicated pass' sums
    form = zkr_hip.sharded_last_form()
    assert form["form"] == "replicated" and "disagreed" in form["reason"] and zkr_hip.sharded_split_stats() is None
    assert "DISAGREED" in capfd.readouterr().err
    monkeypatch.delenv("ZKR_SHARD_SPLIT_CHECK")
    assert zkr_hip.prove_sharded(bad, wb, 61, 62) == want           # remembered: these shards stay replicated
    form = zkr_hip.sharded_last_form()
    assert form["form"] == "replicated" and "first-use check" in form["reason"]
    assert zkr_hip.prove_sharded(good, wb, 61, 62) == want and zkr_hip.sharded_last_form()["form"] == "split"
    monkeypatch.setenv("ZKR_SHARD_SPLIT_H", "1")                    # the opt-in: split whatever the check said
    assert zkr_hip.prove_sharded(bad, wb, 61, 62) == want and zkr_hip.sharded_last_form()["form"] == "split"
    # how a key came to its device
    assert key.replication() == {"mode": "none", "peer_direct": False}
    assert key.replicate(0, "full").replication() == {"mode": "full", "peer_direct": True}
    assert key.replicate(0, "base").replication() == {"mode": "base", "peer_direct": True}


def test_shard_errors_and_memory():
    import zkr_hip
    log_m, p = 12, 73
    key, wb, _ = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF, want_aux=False)
    shards = [key.shard(i, 4) for i in range(4)]
    # the key is sharded, not replicated: the QAP and the twiddles repeat, and a shard's smaller windows mean more levels per point
    # (K = 26 at c = 10 against 22 at c = 12 here)
    assert sum(sh.arena()[1] for sh in shards) < 2 * key.arena()[1]
    assert max(sh.arena()[1] for sh in shards) < 0.5 * key.arena()[1]
    with pytest.raises(zkr_hip.ZkrError, match="itself a shard"):
        shards[1].shard(0, 2)
    # a shard's sums assembled as a proof would be a wrong proof without a sign of it: every proof entry point refuses a shard
    for call in (lambda: shards[0].prove(wb, 1, 2), lambda: shards[0].prove_batch([wb, wb]), lambda: zkr_hip.prove_batch_multi([shards[0], shards[1]], [wb, wb])):
        with pytest.raises(zkr_hip.ZkrError, match="zkr_prove_partial"):
            call()
    with pytest.raises(zkr_hip.ZkrError):
        key.shard(4, 4)
    with pytest.raises(zkr_hip.ZkrError):
        key.shard(0, 65)
    with pytest.raises(zkr_hip.ZkrError, match="position"):
        zkr_hip.prove_sharded([shards[1], shards[0], shards[2], shards[3]], wb, 1, 2)
    with pytest.raises(zkr_hip.ZkrError, match="position"):
        zkr_hip.prove_sharded(shards[:3], wb, 1, 2)
    other, _, _ = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FE, want_aux=False)
    with pytest.raises(zkr_hip.ZkrError, match="another key"):
        zkr_hip.prove_sharded([shards[0], other.shard(1, 4), shards[2], shards[3]], wb, 1, 2)
    with pytest.raises(zkr_hip.ZkrError) as e:
        zkr_hip.prove_sharded(shards, wb[:-32], 1, 2)
    assert e.value.code == -3
    with pytest.raises(zkr_hip.ZkrError):
        shards[0].prove_combine([bytes(zkr_hip.PARTIAL_BYTES)] * 4, g.R, 1)   # blinding out of range
    # a shard is a key: it survives a replication with its ranges (its partial sums are the same group elements; their XYZZ
    # coordinates depend on the order the buckets were filled in, so they are compared through the proof)
    rep = shards[2].replicate(shards[2].device, "base")
    assert rep.shard_info() == shards[2].shard_info()
    parts = [sh.prove_partial(wb) for sh in shards]
    want = key.prove(wb, 5, 6)
    assert key.prove_combine(parts, 5, 6) == want
    assert key.prove_combine(parts[:2] + [rep.prove_partial(wb)] + parts[3:], 5, 6) == want


def test_real_tx_circuit_sharded():
    """The reference's own circuit (tx.circom = BatchProcessTx(2, 6)) set up on the device, its key cut in three."""
    import zkr_hip
    from zkr_hip import rollup
    circ = rollup.RollupCircuit(2, 6)
    key, vk_bin = zkr_hip.ProvingKey.setup_r1cs(circ.r1cs())
    privs = [0x5A4B1000 + 7919 * i for i in range(4)]
    state = rollup.RollupState(circ.depth)
    for i, pv in enumerate(privs):
        state.deposit(i, rollup.gen_public_key(pv), 10 ** 20, 0)
    txs = [state.transfer(j % 4, (j + 1) % 4, 10 ** 17, 10 ** 15, privs[j % 4]) for j in range(circ.batch)]
    wb = circ.calculate_witness(circ.flatten_inputs(state.batch_inputs(txs)))
    shards = [key.shard(i, 3) for i in range(3)]
    assert zkr_hip.prove_sharded(shards, wb, 12345, 67890) == key.prove(wb, 12345, 67890)
    assert zkr_hip.verify(vk_bin, zkr_hip.prove_sharded(shards, wb), circ.public_signals(wb))


def test_keys_come_and_go_without_leaking_device_memory(tmp_path):
    """Replicas, shards (with their rebuilt window levels), packed-file reloads and fresh keys created, used and closed in a loop:
    device memory returns to where it was (tools/key_lifecycle.py is the longer form)."""
    import torch
    import zkr_hip
    log_m = 12
    key, wb, _ = zkr_hip.ProvingKey.synth(log_m, 73, 0x5A4B0001, 0x5A4B00FF, want_aux=False)
    want = key.prove(wb, 3, 4)
    path = str(tmp_path / "k.zkrkey")
    key.save(path)

    def one_round():
        rep = key.replicate(0, "base")
        assert rep.prove(wb, 3, 4) == want
        shards = [key.shard(i, 4) for i in range(4)]
        assert zkr_hip.prove_sharded(shards, wb, 3, 4) == want
        again = zkr_hip.ProvingKey.load_file(path)
        assert again.prove_batch([wb, wb], [3, 3], [4, 4]) == [want, want]
        for k in [rep, again] + shards:
            k.close()

    one_round()                                   # what is built once per process or per key is in place
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(5):
        one_round()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info(0)[0] <= 8 << 20
