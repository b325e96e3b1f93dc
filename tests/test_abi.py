"""CPU: the C-ABI library loads, exports every symbol include/zkr.h declares, and fails loudly without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "zkr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zkr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    import zkr_hip
    L = zkr_hip.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libzkr_hip.so does not export %s" % n


def test_binding_covers_header():
    import zkr_hip.binding as b
    src = open(b.__file__).read()
    for n in _declared():
        assert n in src, "python binding never references %s" % n


def test_no_gpu_means_loud_failure_not_fallback():
    import zkr_hip
    if zkr_hip.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(zkr_hip.ZkrError) as e:
        zkr_hip.ProvingKey.load_websnark(b"\0" * 1000)
    assert e.value.code == -1 and "no CPU fallback" in str(e.value)
    with pytest.raises(zkr_hip.ZkrError):
        zkr_hip.ntt(b"\0" * 64)
    with pytest.raises(zkr_hip.ZkrError):
        zkr_hip.msm_g1(b"\0" * 64, b"\0" * 32)
    with pytest.raises(zkr_hip.ZkrError):
        zkr_hip.ProvingKey.synth(6, 3)


def test_multi_device_and_shard_entry_points_check_their_arguments():
    """The round-4 entry points (zkr_key_replicate, zkr_prove_batch_multi, zkr_key_shard, zkr_prove_partial, zkr_prove_combine,
    zkr_prove_sharded) refuse null arguments with ZKR_ERR_ARG and a message instead of dereferencing them -- checked here without
    a device (the calls return before any HIP work)."""
    import zkr_hip
    L = zkr_hip.lib()
    vp, out, proof = ctypes.c_void_p, ctypes.c_void_p(), ctypes.create_string_buffer(256)
    assert L.zkr_key_replicate(None, 0, 0, ctypes.byref(out)) == -5
    assert L.zkr_key_shard(None, 0, 2, 0, ctypes.byref(out)) == -5
    assert L.zkr_key_device(None) == -1
    assert L.zkr_prove_batch_multi(None, 0, None, 0, 0, None, None, proof) == -5
    assert L.zkr_prove_batch_multi_device(None, 0, None, 0, None, None, proof) == -5
    assert L.zkr_prove_partial(None, b"", 0, proof) == -5 and L.zkr_prove_partial_device(None, None, None, proof) == -5
    assert L.zkr_prove_combine(None, b"", 0, None, None, proof) == -5
    assert L.zkr_prove_sharded(None, 0, b"", 0, None, None, proof) == -5 and L.zkr_prove_sharded_device(None, 0, None, None, None, proof) == -5
    info = (ctypes.c_uint32 * 6)()
    assert L.zkr_key_shard_info(None, info) == -5
    assert b"null" in L.zkr_last_error()
    # round 5: which form a sharded proof took / how a replica came to its device -- nothing ran on this thread yet
    form, mode = ctypes.c_int(7), ctypes.c_int(7)
    assert L.zkr_prove_sharded_last_form(None, None, 0) == -5 and L.zkr_key_replication(None, ctypes.byref(mode), None) == -5
    assert zkr_hip.sharded_last_form() == {"form": "none", "reason": ""}
    buf = ctypes.create_string_buffer(4)
    assert L.zkr_prove_sharded_last_form(ctypes.byref(form), buf, 4) == 0 and form.value == 0 and buf.value == b""


def test_ctypes_batch_calls_refuse_ragged_witnesses_and_short_blinding_lists():
    """the batch wrappers pass ONE witness length and read 32 bytes of r and s per proof -- a shorter witness or a
    shorter rs / ss list must be refused in Python, before the C side reads past the objects (no device needed: the check
    comes first)."""
    import zkr_hip
    from zkr_hip import binding
    k = binding.ProvingKey(ctypes.c_void_p(0), 0)      # never dereferenced: every call below fails its argument check
    k._h = None                                         # ... and its finaliser has nothing to free
    w = b"\0" * 64
    for call in (lambda: zkr_hip.prove_batch_multi([k, k], [w, w[:32]]), lambda: k.prove_batch([w, w + w]),
                 lambda: zkr_hip.prove_batch_multi([k], [w, w], rs=[1], ss=[1, 2]), lambda: zkr_hip.prove_batch_multi([k], [w, w], rs=[1, 2], ss=None),
                 lambda: k.prove_batch([w, w], rs=[1, 2, 3], ss=[1, 2, 3]), lambda: k.prove_batch_device([1, 2], rs=[1], ss=[1]),
                 lambda: zkr_hip.prove_batch_multi_device([k], [1, 2, 3], rs=[1, 2], ss=[1, 2])):
        with pytest.raises(ValueError):
            call()


def test_product_never_imports_oracle():
    """The product path must not route through oracle/: no source under the package mentions it as an import/link."""
    pkg = os.path.join(ROOT, "simple-zk-rollups_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".js", ".c", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"import\s+(coracle|groth16|bn254)|from\s+(coracle|groth16|bn254)|zkr_oracle|libzkr_oracle|require\([^)]*oracle", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_side_witness_generator_matches_oracle_generator():
    """zkr_synth_witness is host-only: product generator == oracle generator, draw for draw."""
    import groth16 as g
    import zkr_hip
    for lm, p, ws in ((5, 3, 9), (9, 7, 0x5A4B0001), (12, 73, 123)):
        wb = zkr_hip.synth_witness(lm, p, 0x5A4B0001, ws)
        c = g.synth_circuit(1 << lm, p, 0x5A4B0001, ws)
        assert g.check_r1cs(c) and g.binarify_witness(c["witness"]) == wb


def test_facade_shapes():
    import zkr_hip
    pb = b"".join(int(i + 1).to_bytes(32, "little") for i in range(8))
    pj = zkr_hip.proof_json_from_bytes(pb)
    assert pj == {"pi_a": ["1", "2", "1"], "pi_b": [["3", "4"], ["5", "6"], ["1", "0"]], "pi_c": ["7", "8", "1"]}
    sp = zkr_hip.solidity_proof(pj, [5, zkr_hip.facade.SNARK_FIELD_SIZE + 2])
    assert sp == {"a": ["1", "2"], "b": [["4", "3"], ["6", "5"]], "c": ["7", "8"], "inputs": ["5", "2"]}


def test_dense_shape_witness_generator_matches_oracle_on_cpu():
    """zkr_synth_witness needs no GPU: the dense-random shape (BASELINE configs[4]) draws the same witness as the
    oracle's generator."""
    import groth16 as g
    import zkr_hip
    zkr_hip.synth_set_shape(1)
    try:
        wb = zkr_hip.synth_witness(7, 5, 0x5A4B0005, 0x5A4B0005)
    finally:
        zkr_hip.synth_set_shape(0)
    circ = g.synth_circuit(128, 5, 0x5A4B0005, shape=1)
    assert g.check_r1cs(circ)
    assert wb == g.binarify_witness(circ["witness"])
    assert zkr_hip.synth_witness(7, 5, 0x5A4B0005, 0x5A4B0005) == g.binarify_witness(g.synth_circuit(128, 5, 0x5A4B0005)["witness"])


def test_native_verifier_agrees_with_oracle_pairing_check(small_case):
    """zkr_verify (host pairing) == the oracle's TxVerifier.sol:258-276 check: accepts the
    closed-form proof, rejects a changed public input, a changed proof element, an input >= r, an off-curve point."""
    import groth16 as g
    import zkr_hip
    c = small_case
    proof = g.proof_from_toxic(c["circ"], c["tox"], c["w"], c["r"], c["s"])
    pb = g.proof_bytes(proof)
    pub = c["w"][1:8]
    vkb = zkr_hip.binarify_verifying_key(c["vk"])
    assert g.is_valid(c["vk"], proof, pub)
    assert zkr_hip.verify(vkb, pb, pub) is True
    assert zkr_hip.is_valid(c["vk"], g.proof_to_json(proof), [str(x) for x in pub]) is True
    bad_pub = list(pub)
    bad_pub[3] = (bad_pub[3] + 1) % g.R
    assert not g.is_valid(c["vk"], proof, bad_pub) and zkr_hip.verify(vkb, pb, bad_pub) is False
    assert zkr_hip.verify(vkb, pb, [pub[0] + g.R] + pub[1:]) is False          # TxVerifier.sol:265: input < r
    other = g.proof_bytes(g.proof_from_toxic(c["circ"], c["tox"], c["w"], c["r"] + 1, c["s"]))
    assert zkr_hip.verify(vkb, other, pub) is True                              # another valid proof of the same statement
    assert zkr_hip.verify(vkb, pb[:192] + other[192:], pub) is False            # pi_c of one with pi_a, pi_b of the other
    assert zkr_hip.verify(vkb, pb[:32] + (1).to_bytes(32, "little") + pb[64:], pub) is False  # pi_a off the curve
    with pytest.raises(zkr_hip.ZkrError):
        zkr_hip.verify(vkb, pb, pub[:-1])                                       # wrong input count (TxVerifier.sol:261)
    with pytest.raises(zkr_hip.ZkrError):
        zkr_hip.verify(vkb[:-1], pb, pub)


def test_native_verifier_accepts_the_reference_contracts_verifying_keys():
    """Golden vectors of the reference: the verifying keys baked into contracts/contracts/TxVerifier.sol:176-257 and
    WithdrawVerifier.sol (tests/golden/verifier_points.json; Solidity limb order [im, re]) are well-formed keys for
    zkr_verify -- every G1 / G2 constant passes its curve check under the product's curve, twist and limb order --
    and moving any coordinate off the curve is refused as a bad key.  (Their proving keys were never committed, so no
    proof can be checked against them.)"""
    import json
    import zkr_hip
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "verifier_points.json")))
    for name, c in fx["contracts"].items():
        n = int(c["n_inputs"])
        g2 = lambda k: ((int(c["g2"][k][0][1]), int(c["g2"][k][0][0])), (int(c["g2"][k][1][1]), int(c["g2"][k][1][0])))  # -> (re, im)
        ic = [c["g1"][k] for k in c["g1"] if k != "alfa1"]
        assert len(ic) == n + 1
        vk = dict(vk_alfa_1=[int(v) for v in c["g1"]["alfa1"]], vk_beta_2=g2("beta2"), vk_gamma_2=g2("gamma2"), vk_delta_2=g2("delta2"),
                  IC=[[int(x), int(y)] for x, y in ic])
        vkb = zkr_hip.binarify_verifying_key(vk)
        some_proof = (1).to_bytes(32, "little") + (2).to_bytes(32, "little") + bytes(192)   # pi_a = G1 generator, rest invalid
        assert zkr_hip.verify(vkb, some_proof, [5] * n) is False          # key accepted, proof (of course) not
        for off in (0, 64 + 32, 64 + 128, 64 + 3 * 128 + 4 + 64 * n):      # alfa1.x, beta2.x.im, gamma2.x.re, last IC x
            bad = bytearray(vkb)
            bad[off] ^= 1
            with pytest.raises(zkr_hip.ZkrError):
                zkr_hip.verify(bytes(bad), some_proof, [5] * n)


def test_batch_verifier_merges_proofs_into_one_pairing_product(small_case):
    """zkr_verify_batch (random linear combination, one final exponentiation) accepts a batch exactly when
    zkr_verify accepts every member: valid proofs of different witnesses pass together, one tampered proof, one wrong
    public signal or one input >= r fails the batch; more pairs than one lock-step Miller loop holds (8) are chunked."""
    import groth16 as g
    import zkr_hip
    c = small_case
    vkb = zkr_hip.binarify_verifying_key(c["vk"])
    proofs, pubs = [], []
    for i in range(7):                                        # 7 + 3 pairs: two Miller-loop chunks
        w = c["w"] if i == 0 else g.synth_circuit(128, 7, 0x5A4B0001, witness_seed=300 + i)["witness"]
        proofs.append(g.proof_bytes(g.proof_from_toxic(c["circ"], c["tox"], w, 11 + i, 23 + i)))
        pubs.append([x % g.R for x in w[1:8]])
    assert all(zkr_hip.verify(vkb, p, s) for p, s in zip(proofs, pubs))
    assert zkr_hip.verify_batch(vkb, proofs, pubs) is True
    assert zkr_hip.verify_batch(vkb, proofs[:1], pubs[:1]) is True and zkr_hip.verify_batch(vkb, [], []) is True
    swapped = [proofs[1], proofs[0]] + proofs[2:]             # every proof valid, but for another statement
    assert zkr_hip.verify_batch(vkb, swapped, pubs) is False
    bad_pub = [list(p) for p in pubs]
    bad_pub[4][2] = (bad_pub[4][2] + 1) % g.R
    assert zkr_hip.verify_batch(vkb, proofs, bad_pub) is False
    big = [list(p) for p in pubs]
    big[3][0] += g.R
    assert zkr_hip.verify_batch(vkb, proofs, big) is False
    off = list(proofs)
    off[5] = off[5][:32] + (1).to_bytes(32, "little") + off[5][64:]
    assert zkr_hip.verify_batch(vkb, off, pubs) is False
    with pytest.raises(zkr_hip.ZkrError):
        zkr_hip.verify_batch(vkb[:-1], proofs, pubs)


def _twist_point_outside_g2():
    """A point on the twist y^2 = x^3 + 3/(9+u) that is NOT in the order-r subgroup, and its multiple by r (a point
    whose order divides the cofactor 2q - r).  Square root in Fq2 by the norm method (q = 3 mod 4)."""
    import bn254 as b
    def sqrt_fq(a):
        s = pow(a, (b.Q + 1) // 4, b.Q)
        return s if s * s % b.Q == a % b.Q else None
    def sqrt_fq2(a):
        a0, a1 = a
        s = sqrt_fq((a0 * a0 + a1 * a1) % b.Q)
        if s is None:
            return None
        for sg in (s, -s):
            x0 = sqrt_fq((a0 + sg) * b.inv(2) % b.Q)
            if x0:
                x1 = a1 * b.inv(2 * x0) % b.Q
                if b.f2sqr((x0, x1)) == (a0 % b.Q, a1 % b.Q):
                    return (x0, x1)
        return None
    k = 1
    while True:
        x = (k, 1)
        y = sqrt_fq2(b.f2add(b.f2mul(b.f2sqr(x), x), b.B2))
        k += 1
        if y is None:
            continue
        P = (x, y)
        assert b.g2_is_on_curve(P)
        Pc = b.g2_mul(P, b.R, reduce=False)
        if Pc is not None:          # [r]P != infinity: P is outside G2 (true for all but a 1/cofactor fraction of points)
            return P, Pc


def test_native_verifier_rejects_twist_points_outside_the_r_torsion(small_case):
    """a G2 point on the twist but outside the order-r subgroup (the twist's cofactor 2q - r has small
    factors) must be refused like the bn256 pairing precompile refuses it (TxVerifier.sol:91-115): as proof.B the
    verdict is False (single and batch), inside a verifying key it is a bad key."""
    import bn254 as b
    import groth16 as g
    import zkr_hip
    c = small_case
    P, Pc = _twist_point_outside_g2()
    assert b.g2_is_on_curve(Pc) and b.g2_mul(Pc, 2 * b.Q - b.R, reduce=False) is None   # order divides the cofactor
    proof = g.proof_from_toxic(c["circ"], c["tox"], c["w"], c["r"], c["s"])
    pb = g.proof_bytes(proof)
    pub = c["w"][1:8]
    vkb = zkr_hip.binarify_verifying_key(c["vk"])
    assert zkr_hip.verify(vkb, pb, pub) is True
    le = lambda v: int(v).to_bytes(32, "little")
    for bad in (P, Pc):
        enc = le(bad[0][0]) + le(bad[0][1]) + le(bad[1][0]) + le(bad[1][1])
        forged = pb[:64] + enc + pb[192:]
        assert zkr_hip.verify(vkb, forged, pub) is False
        assert zkr_hip.verify_batch(vkb, [pb, forged], [pub, pub]) is False
        for off in (64, 192, 320):                            # beta2, gamma2, delta2 of the key
            with pytest.raises(zkr_hip.ZkrError):
                zkr_hip.verify(vkb[:off] + enc + vkb[off + 128:], pb, pub)
    # the subgroup's own points still pass: the key's gamma2 used as proof.B is on the twist AND in G2 (verdict False, no error)
    assert zkr_hip.verify(vkb, pb[:64] + vkb[192:320] + pb[192:], pub) is False


def test_solidity_verifier_constants_round_trip_the_reference_contracts():
    """SURVEY 8(f-2) tail, pinned on the reference's own vectors both ways: the verifying-key constants of
    contracts/contracts/TxVerifier.sol:177-255 and WithdrawVerifier.sol (tests/golden/verifier_points.json, Solidity limb
    order [im, re]) -> vk_bin -> solidity_verifying_key() gives back exactly those constants, and the emitted
    `verifyingKey()` statements parse back to them too (the regular expressions of tests/golden/make_verifier_points.py)."""
    import json
    import zkr_hip
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "verifier_points.json")))
    for name, c in fx["contracts"].items():
        n = int(c["n_inputs"])
        to_js = lambda k: ((int(c["g2"][k][0][1]), int(c["g2"][k][0][0])), (int(c["g2"][k][1][1]), int(c["g2"][k][1][0])))  # [im, re] -> (re, im)
        ic_names = ["IC[%d]" % i for i in range(n + 1)]
        vk = dict(vk_alfa_1=[int(v) for v in c["g1"]["alfa1"]], vk_beta_2=to_js("beta2"), vk_gamma_2=to_js("gamma2"), vk_delta_2=to_js("delta2"),
                  IC=[[int(v) for v in c["g1"][k]] for k in ic_names])
        vkb = zkr_hip.binarify_verifying_key(vk)
        sol = zkr_hip.solidity_verifying_key(vkb)
        assert sol["alfa1"] == c["g1"]["alfa1"] and sol["IC"] == [c["g1"][k] for k in ic_names]
        for k in ("beta2", "gamma2", "delta2"):
            assert sol[k] == c["g2"][k], (name, k)
        # and through the JSON form the product hands to the reference's tooling
        assert zkr_hip.binarify_verifying_key(zkr_hip.verifying_key_from_bytes(vkb)) == vkb
        src = zkr_hip.solidity_verifying_key_source(vkb)
        g1 = re.findall(r"vk\.(alfa1|IC\[\d+\]) = Pairing\.G1Point\((\d+),\s*(\d+)\)", src)
        g2 = re.findall(r"vk\.(beta2|gamma2|delta2) = Pairing\.G2Point\(\[(\d+),\s*(\d+)\], \[(\d+),\s*(\d+)\]\)", src)
        assert {k: [x, y] for k, x, y in g1} == c["g1"]
        assert {k: [[a, b], [cc, d]] for k, a, b, cc, d in g2} == c["g2"]
        assert "vk.IC = new Pairing.G1Point[](%d);" % (n + 1) in src
    with pytest.raises(ValueError):
        zkr_hip.solidity_verifying_key(vkb[:-1])
