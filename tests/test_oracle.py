"""CPU: pin the oracle.  (a) the reference's in-repo constants (verifier contracts) decode under the
oracle's curve equations, (b) independent prover routes agree with the toxic-waste closed form and the
TxVerifier.sol pairing equation, (c) the C restatement agrees with the Python one, (d) golden fixtures."""
import json
import os
import random

import pytest

import bn254 as bn
import coracle
import groth16 as g
from bn254 import Q, R

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _le(v):
    return int(v).to_bytes(32, "little")


# ---------------------------------------------------------------- (a) constants from the reference tree
def test_moduli_and_generators_match_reference_constants():
    # binarify.ts:79-81,86-88 ; TxVerifier.sol:24-35,50,259
    assert Q == int("21888242871839275222246405745257275088696311157297823662689037894645226208583")
    assert R == int("21888242871839275222246405745257275088548364400416034343698204186575808495617")
    fx = json.load(open(os.path.join(GOLD, "verifier_points.json")))
    gen = fx["contracts"]["TxVerifier"]["g2_generator"]  # Solidity order [im, re]
    assert ((int(gen[0][1]), int(gen[0][0])), (int(gen[1][1]), int(gen[1][0]))) == bn.G2_GEN
    assert bn.g1_is_on_curve(bn.G1_GEN) and bn.g2_is_on_curve(bn.G2_GEN)
    assert bn.g2_mul(bn.G2_GEN, R, reduce=False) is None


def test_reference_verifier_constants_are_valid_group_elements():
    """80 G1 + 6 G2 constants of TxVerifier.sol:176-257 / WithdrawVerifier.sol: on-curve, G2 in the r-torsion."""
    fx = json.load(open(os.path.join(GOLD, "verifier_points.json")))
    n1 = n2 = 0
    for name, c in fx["contracts"].items():
        assert len(c["g1"]) == c["n_inputs"] + 2  # alfa1 + IC[0..n]
        for k, (x, y) in c["g1"].items():
            assert bn.g1_is_on_curve((int(x), int(y))), (name, k)
            n1 += 1
        for k, ((x1, x0), (y1, y0)) in c["g2"].items():
            P = ((int(x0), int(x1)), (int(y0), int(y1)))
            assert bn.g2_is_on_curve(P), (name, k)
            assert bn.g2_mul(P, R, reduce=False) is None, (name, k)
            n2 += 1
    assert (n1, n2) == (80, 6)


def test_pairing_bilinear_and_nondegenerate():
    e1 = bn.pairing(bn.G2_GEN, bn.G1_GEN)
    assert e1 != bn.F12_ONE and bn.f12pow(e1, R) == bn.F12_ONE
    a, b = 0x1234567, 0x89ABCDEF01
    assert bn.pairing(bn.g2_mul(bn.G2_GEN, b), bn.g1_mul(bn.G1_GEN, a)) == bn.f12pow(e1, a * b)


# ---------------------------------------------------------------- (b) prover routes vs closed form vs pairing
@pytest.fixture(scope="module")
def tiny():
    circ = g.synth_circuit(32, 3, 0x5A4B0001)
    tox = g.toxic_from_seed(0x5A4B00FF)
    pk, vk = g.setup(circ, tox)
    return circ, tox, pk, vk


def test_synthetic_circuit_is_satisfied_and_rollup_shaped():
    circ = g.synth_circuit(4096, 73, 0x5A4B0001)
    assert g.check_r1cs(circ)
    assert circ["nConstraints"] + 73 + 1 == 4096 == g.domain_size(circ["nConstraints"], 73)
    w = circ["witness"]
    frac_bits = sum(1 for x in w if x in (0, 1)) / len(w)
    assert 0.01 < frac_bits < 0.06
    nnz = sum(len(A) + len(B) for A, B, C in circ["rows"]) / len(circ["rows"])
    assert 2.0 < nnz < 5.0
    w2 = g.synth_circuit(4096, 73, 0x5A4B0001, witness_seed=7)
    assert g.check_r1cs(w2) and w2["rows"] == circ["rows"] and w2["witness"] != w


def test_three_h_routes_and_closed_form_agree_and_verify(tiny):
    circ, tox, pk, vk = tiny
    w = circ["witness"]
    rng = g.SplitMix64(5)
    r, s = rng.fr(), rng.fr()
    h = g.calc_h_snarkjs(pk, w)
    assert h == g.calc_h_websnark(pk, w) == g.calc_h_halves(pk, w)
    p1 = g.prove(pk, w, r, s, "snarkjs")
    assert p1 == g.prove(pk, w, r, s, "halves") == g.proof_from_toxic(circ, tox, w, r, s)
    assert p1 == g.proof_from_toxic(circ, tox, w, r, s, h=h)
    pub = w[1:4]
    assert g.is_valid(vk, p1, pub)
    assert not g.is_valid(vk, p1, [pub[0], (pub[1] + 1) % R, pub[2]])  # withdrawverifier.test.ts:42-68
    assert not g.is_valid(vk, p1, [pub[0], pub[1] + R, pub[2]])        # TxVerifier.sol:265 bound
    p2 = g.prove(pk, w, (r + 1) % R, s, "websnark")
    assert p2 != p1 and g.is_valid(vk, p2, pub)


def test_unsatisfied_witness_yields_invalid_proof_on_every_route(tiny):
    circ, tox, pk, vk = tiny
    w = list(circ["witness"])
    w[9] = (w[9] + 1) % R
    q = g.prove(pk, w, 11, 22, "snarkjs")
    assert q == g.prove(pk, w, 11, 22, "websnark") == g.prove(pk, w, 11, 22, "halves")
    assert not g.is_valid(vk, q, w[1:4])


def test_binarify_layouts(tiny):
    circ, tox, pk, vk = tiny
    n, p, m = pk["nVars"], pk["nPublic"], pk["domainSize"]
    wb = g.binarify_witness(circ["witness"])
    assert len(wb) == 32 * n and wb[:32] == _le(1)                   # binarify.ts:28-30, witness[0] = 1
    pkb = g.binarify_proving_key(g.to_json_key(pk))
    size = 40 + 192 + 256 + sum(36 * len(d) + 4 for d in pk["polsA"]) + sum(36 * len(d) + 4 for d in pk["polsB"]) \
        + 64 * n * 2 + 128 * n + 64 * (n - p - 1) + 64 * m          # binarify.ts:115-141
    assert len(pkb) == size
    u32 = lambda o: int.from_bytes(pkb[o:o + 4], "little")
    assert (u32(0), u32(4), u32(8), u32(12)) == (n, p, m, 488)
    assert pkb[40:72] == _le(pk["vk_alfa_1"][0] * (1 << 256) % Q)   # toMontgomeryQ, binarify.ts:78-83
    back = g.parse_proving_key(pkb)
    for f in ("A", "B1", "B2", "polsA", "polsB", "vk_beta_2", "vk_delta_1"):
        assert back[f] == pk[f], f
    assert back["hExps"] == pk["hExps"][:m] and back["C"][p + 1:] == pk["C"][p + 1:]
    assert any(P is None for P in pk["B1"])                          # infinity = (0, mont(1)) round-trips
    pj = g.proof_to_json(g.proof_from_toxic(circ, tox, circ["witness"], 3, 4))
    sp = g.solidity_proof(pj, circ["witness"][1:p + 1])
    assert sp["b"][0] == [pj["pi_b"][0][1], pj["pi_b"][0][0]] and len(sp["a"]) == 2 and len(sp["inputs"]) == p  # common.ts:43-50


# ---------------------------------------------------------------- (c) C restatement == Python restatement
def test_c_oracle_field_and_ntt():
    import ctypes
    rnd = random.Random(1)
    for f, P in ((coracle.lib().zo_fq_mul_std, Q), (coracle.lib().zo_fr_mul_std, R)):
        for _ in range(100):
            a, b = rnd.randrange(P), rnd.randrange(P)
            o = ctypes.create_string_buffer(32)
            f(_le(a), _le(b), o)
            assert int.from_bytes(o.raw, "little") == a * b % P
    for logn in (1, 3, 8):
        v = [rnd.randrange(R) for _ in range(1 << logn)]
        d = b"".join(_le(x) for x in v)
        assert coracle.ntt(d) == b"".join(_le(x) for x in g.ntt(v))
        assert coracle.ntt(coracle.ntt(d), inverse=True) == d


def test_c_oracle_msm_calc_h_prove(small_case):
    c = small_case
    pk = c["pk"]
    mont = lambda v: _le(v * (1 << 256) % Q)
    pts = b"".join(_le(0) + mont(1) if P is None else mont(P[0]) + mont(P[1]) for P in pk["A"])
    got = coracle.msm_g1(pts, c["wb"])
    exp = bn.g1_msm(pk["A"], c["w"])
    assert got == _le(exp[0]) + _le(exp[1])
    h = g.calc_h_websnark(pk, c["w"])
    assert coracle.calc_h(c["pkb"], c["wb"]) == b"".join(_le(x) for x in h)
    proof = coracle.prove(c["pkb"], c["wb"], c["r"], c["s"])
    assert proof == g.proof_bytes(g.proof_from_toxic(c["circ"], c["tox"], c["w"], c["r"], c["s"]))


# ---------------------------------------------------------------- (d) golden fixtures
def test_c_oracle_all_threads_prover_and_dot_product(small_case):
    """bench.py's all-host-threads CPU baseline (zo_prove_mt: OpenMP slices of the five multiexps, parallel NTT loops)
    returns the bytes of the single-thread prover for every thread count, also on the golden fixture sizes; zo_fr_dot
    (the closed form's dot products at 2^24) equals the Python sum."""
    c = small_case
    want = coracle.prove(c["pkb"], c["wb"], c["r"], c["s"])
    for threads in (1, 2, 3, 8, 0):
        got, tm = coracle.prove_mt(c["pkb"], c["wb"], c["r"], c["s"], threads=threads, want_timings=True)
        assert got == want and tm[3] >= 1
    assert coracle.max_threads() >= 1
    rng = g.SplitMix64(31)
    a = [rng.fr() for _ in range(1000)] + [0, R - 1, 1]
    b = [rng.fr() for _ in range(1000)] + [5, R - 1, R - 1]
    assert coracle.fr_dot(b"".join(_le(x) for x in a), b"".join(_le(x) for x in b)) == sum(x * y for x, y in zip(a, b)) % R
    assert coracle.fr_dot(b"", b"") == 0


def test_cpu_baseline_build_choice_keeps_the_oracles_results(tmp_path, small_case):
    """bench.py's CPU baseline compiles the C oracle with `-march=native` on the box and times whichever of the native and the
    portable (x86-64-v2) build is faster there (coracle.use_fastest; BASELINE.md section 3 promises a native build): whichever is
    chosen must produce the portable build's bytes.  Its own process: the choice is made once, before the library is loaded."""
    import pickle
    import subprocess
    import sys
    c = small_case
    want = coracle.prove(c["pkb"], c["wb"], c["r"], c["s"])
    case = tmp_path / "case.pkl"
    case.write_bytes(pickle.dumps((c["pkb"], c["wb"], c["r"], c["s"])))
    script = (
        "import sys, pickle; sys.path.insert(0, %r); import coracle\n"
        "pk, w, r, s = pickle.load(open(%r, 'rb'))\n"
        "how = coracle.use_fastest(pk, w)\n"
        "print('HOW', how)\n"
        "sys.stdout.write('PROOF ' + coracle.prove(pk, w, r, s).hex() + '\\n')\n"
        "print('AGAIN', coracle.use_fastest(pk, w))\n" % (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"), str(case)))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = dict(ln.split(" ", 1) for ln in r.stdout.splitlines() if " " in ln)
    assert lines["PROOF"] == want.hex()
    assert "march=" in lines["HOW"] and ("faster on this host" in lines["HOW"] or "portable build" in lines["HOW"])
    assert lines["AGAIN"].startswith("already loaded")          # the choice cannot change under a loaded library


@pytest.mark.parametrize("name", ["synth_m5.json", "synth_m7.json"])
def test_golden_fixture_reproduces(name):
    fx = json.load(open(os.path.join(GOLD, name)))
    circ = g.synth_circuit(1 << fx["log_m"], fx["n_public"], fx["circuit_seed"])
    tox = g.toxic_from_seed(fx["toxic_seed"])
    pk, vk = g.setup(circ, tox)
    pkb = g.binarify_proving_key(g.to_json_key(pk))
    wb = g.binarify_witness(circ["witness"])
    assert (g.sha256(pkb), len(pkb), g.sha256(wb)) == (fx["pk_bin_sha256"], fx["pk_bin_len"], fx["witness_bin_sha256"])
    h = coracle.calc_h(pkb, wb)
    assert g.sha256(h) == fx["h_sha256"]
    assert [str(int.from_bytes(h[32 * i:32 * i + 32], "little")) for i in range(4)] == fx["h_first4"]
    proof = coracle.prove(pkb, wb, int(fx["r"]), int(fx["s"]))
    assert proof.hex() == fx["proof_bytes_hex"]
    assert fx["is_valid"] is True


def _stringify(x):
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return str(x)
    if isinstance(x, dict):
        return {str(k): _stringify(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_stringify(v) for v in x]
    return x


def test_js_snarkjs_style_baseline_equals_closed_form(tmp_path, small_case):
    """oracle/cpu_ref_js.js (single-thread snarkjs-0.1.20-shaped genProof on native BigInt: the CPU baseline of
    bench.py) produces the closed-form proof, so the time it reports is for the right computation."""
    import json
    import shutil
    import subprocess
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not available")
    c = small_case
    jk = _stringify(g.to_json_key(c["pk"]))
    for f in ("nVars", "nPublic", "domainSize", "domainBits"):
        jk[f] = int(jk[f])
    path = tmp_path / "case.json"
    path.write_text(json.dumps(dict(pk=jk, witness=[str(x) for x in c["w"]], r=str(c["r"]), s=str(c["s"]))))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([node, os.path.join(root, "oracle", "cpu_ref_js.js"), str(path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout)
    assert res["proof"] == g.proof_to_json(g.proof_from_toxic(c["circ"], c["tox"], c["w"], c["r"], c["s"]))
    assert res["seconds"]["total"] > 0
