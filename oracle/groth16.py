"""Groth16 setup / prove / verify + boundary codecs -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
PARITY UNPINNED for proof bytes (see oracle/bn254.py header and SURVEY.md 8(c)): the
reference's tests assert validity only, inputs and blinding are random each run.

What is restated here and what it follows:
  binarify_witness / binarify_proving_key   /root/reference/operator/src/utils/binarify.ts:10-48, 50-207
      (byte layouts are authoritative and in-repo)
  setup()        snarkjs@0.1.20 `setup --protocol groth` (prover/package.json:34,37; dependency
                 pinned at prover/yarn.lock:4832-4842, un-vendored)            [published algorithm]
  prove_snarkjs()  snarkjs@0.1.20 groth.genProof: per-signal scalar muls, calculateH via
                 coefficient product                                           [published algorithm]
  prove_websnark() websnark@0.0.5 groth16GenProof (call site operator/src/snarks/common.ts:29):
                 h through the 2m-domain NTT route, then five multiexps        [published algorithm]
  is_valid()     snarkjs groth.isValid == contracts/contracts/TxVerifier.sol:258-276
  solidity_proof()  operator/src/snarks/common.ts:40-51
Both prove_* routes and the toxic-waste closed form (proof_from_toxic) must agree bit-for-bit.
"""
import hashlib

from bn254 import (Q, R, G1_GEN, G2_GEN, g1_add, g1_mul, g1_neg, g2_add, g2_mul, g1_msm, g2_msm,
                   pairing_product_is_one, inv, _jdbl, _jadd_mixed, _jaffine)

MONT = 1 << 256


# ----------------------------------------------------------------------------- PRNG / workload
class SplitMix64:
    """Deterministic stream shared with the product-side generator (csrc/workload.cpp)."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self):
        while True:
            v = self.u64() | (self.u64() << 64) | (self.u64() << 128) | ((self.u64() & ((1 << 62) - 1)) << 192)
            if v < R:
                return v


def synth_circuit(m, p, seed, witness_seed=None, shape=0):
    """Rollup-shaped synthetic R1CS with witness by forward evaluation (SURVEY.md 8(d) config 2,
    Appendix D): ~93 % MiMC-like multiplication rows with 1-3 nnz per A/B/C row, 3 % boolean
    rows, 2 % small (64-bit) values, and a 64-term packing row every 2048 rows.
    Structure (row kinds, wiring, coefficients) is drawn from SplitMix64(seed); free witness values
    (public inputs, booleans, small values) from SplitMix64(witness_seed ^ 0x77697473), so one key
    serves a batch of different witnesses.  witness_seed defaults to seed.
    shape=1 (BASELINE config 5, "dense random"): every row is (sum of 4 random signals with random coefficients) x
    (another such sum) = new signal, so every signal has A-, B1- and B2-query points (no infinities: G2 stress) and
    the QAP rows gather from uniformly random columns.
    Returns dict(nVars, nPublic, nConstraints, rows=[(A,B,C)], witness=[...]); A/B/C are
    lists of (signal, coef).  Mirrors synth_circuit in csrc/workload.hip draw for draw."""
    rng = SplitMix64(seed)
    rv = SplitMix64((seed if witness_seed is None else witness_seed) ^ 0x77697473)
    nC = m - p - 1
    w = [1] + [rv.fr() for _ in range(p)]
    rows = []
    for c in range(nC):
        n = len(w)
        if shape == 1:
            A, B = {}, {}
            for d in (A, B):
                for _ in range(4):
                    j = rng.u64() % n
                    d[j] = (d.get(j, 0) + rng.fr()) % R
            va = sum(cf * w[s] for s, cf in A.items()) % R
            vb = sum(cf * w[s] for s, cf in B.items()) % R
            w.append(va * vb % R)
            rows.append((sorted(A.items()), sorted(B.items()), [(n, 1)]))
            continue
        kind = rng.u64() % 100
        if c % 2048 == 1000:
            A = {}
            acc = 0
            for k in range(64):
                j = rng.u64() % n
                A[j] = (A.get(j, 0) + (1 << k)) % R
            for j, cf in A.items():
                acc = (acc + cf * w[j]) % R
            new = n
            w.append(acc)
            rows.append((sorted(A.items()), [(0, 1)], [(new, 1)]))
        elif kind < 3:
            b = rv.u64() & 1
            new = n
            w.append(b)
            rows.append(([(new, 1)], [(0, R - 1), (new, 1)], []))
        elif kind < 5:
            v = rv.u64()
            new = n
            w.append(v)
            rows.append(([(new, 1)], [(0, 1)], [(new, 1)]))
        else:
            i = n - 1
            sel = rng.u64()
            j = rng.u64() % n
            kk = n - 1 - (rng.u64() % min(n - 1, 16)) if n > 1 else 0
            A = {i: 1}
            if sel & 1 and j != i:
                A[j] = rng.fr()
            if (sel >> 1) & 3 == 0 and 0 not in A:
                A[0] = rng.fr()
            B = {kk: 1}
            if (sel >> 3) & 1 and kk != 0:
                B[0] = rng.fr()
            va = sum(cf * w[s] for s, cf in A.items()) % R
            vb = sum(cf * w[s] for s, cf in B.items()) % R
            new = n
            C = [(new, 1)]
            val = va * vb % R
            if (sel >> 4) & 3 == 0:
                i2 = rng.u64() % n
                C = [(i2, R - 1), (new, 1)]
                val = (val + w[i2]) % R
            w.append(val)
            rows.append((sorted(A.items()), sorted(B.items()), C))
    return dict(nVars=len(w), nPublic=p, nConstraints=nC, domainSize=m, rows=rows, witness=w)


def check_r1cs(circ):
    w = circ["witness"]
    for A, B, C in circ["rows"]:
        a = sum(cf * w[s] for s, cf in A) % R
        b = sum(cf * w[s] for s, cf in B) % R
        c = sum(cf * w[s] for s, cf in C) % R
        if a * b % R != c:
            return False
    return True


# ----------------------------------------------------------------------------- NTT over Fr
def root_of_unity(n):
    """omega_n = 5^((r-1)/n): 5 is the smallest quadratic non-residue mod r."""
    assert n & (n - 1) == 0 and (R - 1) % n == 0
    return pow(5, (R - 1) // n, R)


def ntt(a, invert=False):
    n = len(a)
    a = list(a)
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j ^= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = root_of_unity(length)
        if invert:
            wl = inv(wl, R)
        half = length >> 1
        tw = [1] * half
        for k in range(1, half):
            tw[k] = tw[k - 1] * wl % R
        for i in range(0, n, length):
            for k in range(half):
                u = a[i + k]
                v = a[i + k + half] * tw[k] % R
                a[i + k] = (u + v) % R
                a[i + k + half] = (u - v) % R
        length <<= 1
    if invert:
        ni = inv(n, R)
        a = [x * ni % R for x in a]
    return a


def lagrange_at(m, t):
    """[L_c(t)] for the domain {omega^c}: L_c(t) = (t^m - 1) omega^c / (m (t - omega^c))."""
    w = root_of_unity(m)
    z = (pow(t, m, R) - 1) % R
    mi = inv(m, R)
    out = []
    wc = 1
    for _ in range(m):
        out.append(z * wc % R * mi % R * inv((t - wc) % R, R) % R)
        wc = wc * w % R
    return out


# ----------------------------------------------------------------------------- fixed-base tables
class _FixedBaseG1:
    def __init__(self, P):
        self.tbl = []
        cur = P
        for _ in range(254):
            self.tbl.append(cur)
            cur = g1_add(cur, cur)

    def mul(self, k):
        k %= R
        acc = (1, 1, 0)
        i = 0
        while k:
            if k & 1:
                acc = _jadd_mixed(*acc, *self.tbl[i])
            k >>= 1
            i += 1
        return _jaffine(*acc)


class _FixedBaseG2:
    def __init__(self, P):
        self.tbl = []
        cur = P
        for _ in range(254):
            self.tbl.append(cur)
            cur = g2_add(cur, cur)

    def mul(self, k):
        k %= R
        acc = None
        i = 0
        while k:
            if k & 1:
                acc = g2_add(acc, self.tbl[i])
            k >>= 1
            i += 1
        return acc


_FB1 = None
_FB2 = None


def _fb():
    global _FB1, _FB2
    if _FB1 is None:
        _FB1 = _FixedBaseG1(G1_GEN)
        _FB2 = _FixedBaseG2(G2_GEN)
    return _FB1, _FB2


# ----------------------------------------------------------------------------- setup
def toxic_from_seed(seed):
    rng = SplitMix64(seed)
    return dict(t=rng.fr(), alfa=rng.fr(), beta=rng.fr(), gamma=rng.fr(), delta=rng.fr())


def qap_columns(circ):
    """polsA/polsB/polsC per signal as {constraint: coef}, including the nPublic+1
    input-consistency rows polsA[i][nC+i] = 1 that snarkjs setup appends."""
    n, p, nC = circ["nVars"], circ["nPublic"], circ["nConstraints"]
    polsA = [dict() for _ in range(n)]
    polsB = [dict() for _ in range(n)]
    polsC = [dict() for _ in range(n)]
    for c, (A, B, C) in enumerate(circ["rows"]):
        for s, cf in A:
            polsA[s][c] = cf
        for s, cf in B:
            polsB[s][c] = cf
        for s, cf in C:
            polsC[s][c] = cf
    for i in range(p + 1):
        polsA[i][nC + i] = 1
    return polsA, polsB, polsC


def domain_size(nC, p):
    total = nC + p + 1
    bits = (total - 1).bit_length()  # floor(log2(total-1)) + 1
    return 1 << bits


def setup_scalars(circ, tox):
    """Discrete logs of every key element (what the toxic waste knows)."""
    n, p, nC = circ["nVars"], circ["nPublic"], circ["nConstraints"]
    m = domain_size(nC, p)
    assert m == circ["domainSize"]
    polsA, polsB, polsC = qap_columns(circ)
    L = lagrange_at(m, tox["t"])
    a = [sum(cf * L[c] for c, cf in polsA[s].items()) % R for s in range(n)]
    b = [sum(cf * L[c] for c, cf in polsB[s].items()) % R for s in range(n)]
    cc = [sum(cf * L[c] for c, cf in polsC[s].items()) % R for s in range(n)]
    ginv = inv(tox["gamma"], R)
    dinv = inv(tox["delta"], R)
    k = [(tox["beta"] * a[s] + tox["alfa"] * b[s] + cc[s]) % R for s in range(n)]
    z = (pow(tox["t"], m, R) - 1) % R
    zd = z * dinv % R
    hx = []
    ti = 1
    for _ in range(m + 1):
        hx.append(ti * zd % R)
        ti = ti * tox["t"] % R
    return dict(m=m, a=a, b=b, c=cc, ic=[k[s] * ginv % R for s in range(p + 1)],
                cpriv=[None if s <= p else k[s] * dinv % R for s in range(n)], h=hx,
                polsA=polsA, polsB=polsB, polsC=polsC)


def setup(circ, tox):
    """-> (proving key dict in the snarkjs JSON schema read by binarify.ts:129-202, verifying key dict).
    Points are affine tuples / None here; to_json_key() renders the JSON shapes."""
    sc = setup_scalars(circ, tox)
    fb1, fb2 = _fb()
    n, p = circ["nVars"], circ["nPublic"]
    pk = dict(protocol="groth", nVars=n, nPublic=p, domainSize=sc["m"], domainBits=sc["m"].bit_length() - 1,
              polsA=sc["polsA"], polsB=sc["polsB"], polsC=sc["polsC"],
              A=[fb1.mul(x) for x in sc["a"]], B1=[fb1.mul(x) for x in sc["b"]],
              B2=[fb2.mul(x) for x in sc["b"]],
              C=[None if x is None else fb1.mul(x) for x in sc["cpriv"]],
              hExps=[fb1.mul(x) for x in sc["h"]],
              vk_alfa_1=fb1.mul(tox["alfa"]), vk_beta_1=fb1.mul(tox["beta"]), vk_delta_1=fb1.mul(tox["delta"]),
              vk_beta_2=fb2.mul(tox["beta"]), vk_delta_2=fb2.mul(tox["delta"]))
    vk = dict(protocol="groth", nPublic=p, IC=[fb1.mul(x) for x in sc["ic"]],
              vk_alfa_1=pk["vk_alfa_1"], vk_beta_2=pk["vk_beta_2"],
              vk_gamma_2=fb2.mul(tox["gamma"]), vk_delta_2=pk["vk_delta_2"])
    return pk, vk


# ----------------------------------------------------------------------------- prove
def qap_evaluate(pk, witness):
    m = pk["domainSize"]
    a = [0] * m
    b = [0] * m
    for s in range(pk["nVars"]):
        ws = witness[s]
        if ws == 0:
            continue
        for c, cf in pk["polsA"][s].items():
            a[c] = (a[c] + cf * ws) % R
        for c, cf in pk["polsB"][s].items():
            b[c] = (b[c] + cf * ws) % R
    return a, b


def calc_h_snarkjs(pk, witness):
    """snarkjs calculateH: coefficient-domain product, upper half (C has degree < m so it does
    not reach the upper half; polsC is not needed -- SURVEY App. B step 3)."""
    m = pk["domainSize"]
    a, b = qap_evaluate(pk, witness)
    ac = ntt(a, invert=True)
    bc = ntt(b, invert=True)
    prod = ntt([x * y % R for x, y in zip(ntt(ac + [0] * m), ntt(bc + [0] * m))], invert=True)
    return prod[m:]


def calc_h_websnark(pk, witness):
    """websnark route: evaluate A,B on the odd 2m-th roots (coset omega_2m), interleave with the
    domain evaluations, pointwise multiply, iNTT_2m, upper half."""
    m = pk["domainSize"]
    a, b = qap_evaluate(pk, witness)
    ac = ntt(a, invert=True)
    bc = ntt(b, invert=True)
    g = root_of_unity(2 * m)
    gi = 1
    acs, bcs = [], []
    for i in range(m):
        acs.append(ac[i] * gi % R)
        bcs.append(bc[i] * gi % R)
        gi = gi * g % R
    ao = ntt(acs)
    bo = ntt(bcs)
    ev = [0] * (2 * m)
    for c in range(m):
        ev[2 * c] = a[c] * b[c] % R
        ev[2 * c + 1] = ao[c] * bo[c] % R
    return ntt(ev, invert=True)[m:]


def calc_h_halves(pk, witness):
    """The decomposition the HIP path uses: with P = A*B = P_lo + x^m P_hi,
    S = P mod (x^m - 1) = iNTT_m(a.b), D = P mod (x^m + 1) = g^-i . iNTT_m(A(g w^c) B(g w^c)),
    h = P_hi = (S - D)/2.  Exact for every witness, satisfying or not."""
    m = pk["domainSize"]
    a, b = qap_evaluate(pk, witness)
    ac = ntt(a, invert=True)
    bc = ntt(b, invert=True)
    g = root_of_unity(2 * m)
    gp = [1] * m
    for i in range(1, m):
        gp[i] = gp[i - 1] * g % R
    ao = ntt([x * y % R for x, y in zip(ac, gp)])
    bo = ntt([x * y % R for x, y in zip(bc, gp)])
    S = ntt([x * y % R for x, y in zip(a, b)], invert=True)
    Dg = ntt([x * y % R for x, y in zip(ao, bo)], invert=True)
    ginv = inv(g, R)
    half = inv(2, R)
    out = []
    gi = 1
    for i in range(m):
        out.append((S[i] - Dg[i] * gi) * half % R)
        gi = gi * ginv % R
    return out


def _assemble(pk, A_msm, B1_msm, B2_msm, C_msm, H_msm, r, s):
    pi_a = g1_add(g1_add(A_msm, pk["vk_alfa_1"]), g1_mul(pk["vk_delta_1"], r))
    pi_b = g2_add(g2_add(B2_msm, pk["vk_beta_2"]), g2_mul(pk["vk_delta_2"], s))
    pib1 = g1_add(g1_add(B1_msm, pk["vk_beta_1"]), g1_mul(pk["vk_delta_1"], s))
    pi_c = g1_add(C_msm, H_msm)
    pi_c = g1_add(pi_c, g1_mul(pi_a, s))
    pi_c = g1_add(pi_c, g1_mul(pib1, r))
    pi_c = g1_add(pi_c, g1_mul(pk["vk_delta_1"], (-(r * s)) % R))
    return dict(pi_a=pi_a, pi_b=pi_b, pi_c=pi_c)


def prove(pk, witness, r, s, route="websnark"):
    """-> dict(pi_a, pi_b, pi_c) as affine points.  route in {'snarkjs','websnark','halves'}: only
    the h computation differs; results must be identical."""
    n, p, m = pk["nVars"], pk["nPublic"], pk["domainSize"]
    w = [x % R for x in witness]
    h = dict(snarkjs=calc_h_snarkjs, websnark=calc_h_websnark, halves=calc_h_halves)[route](pk, w)
    A_msm = g1_msm(pk["A"], w)
    B1_msm = g1_msm(pk["B1"], w)
    B2_msm = g2_msm(pk["B2"], w)
    C_msm = g1_msm(pk["C"][p + 1:], w[p + 1:])
    H_msm = g1_msm(pk["hExps"][:m], h)
    return _assemble(pk, A_msm, B1_msm, B2_msm, C_msm, H_msm, r, s)


def proof_from_toxic(circ, tox, witness, r, s, h=None):
    """Closed form with the toxic waste known: no MSM, no NTT needed except
    for h (pass it in, or it is derived as (A.B - C)/Z evaluated at t for a satisfying witness)."""
    sc = setup_scalars(circ, tox)
    n, p, m = circ["nVars"], circ["nPublic"], sc["m"]
    w = [x % R for x in witness]
    At = sum(w[i] * sc["a"][i] for i in range(n)) % R
    Bt = sum(w[i] * sc["b"][i] for i in range(n)) % R
    Ct = sum(w[i] * sc["c"][i] for i in range(n)) % R
    d = tox["delta"]
    a_log = (tox["alfa"] + At + r * d) % R
    b_log = (tox["beta"] + Bt + s * d) % R
    if h is None:
        z = (pow(tox["t"], m, R) - 1) % R
        hz_over_d = (At * Bt - Ct) % R * inv(d, R) % R  # h(t) Z(t) / delta
        del z
    else:
        hz_over_d = sum(hi * x for hi, x in zip(h, sc["h"])) % R
    cpriv = sum(w[i] * sc["cpriv"][i] for i in range(p + 1, n)) % R
    c_log = (cpriv + hz_over_d + s * a_log + r * b_log - r * s % R * d) % R
    fb1, fb2 = _fb()
    return dict(pi_a=fb1.mul(a_log), pi_b=fb2.mul(b_log), pi_c=fb1.mul(c_log))


# ----------------------------------------------------------------------------- verify
def is_valid(vk, proof, public_signals):
    """e(A,B) == e(alfa,beta) e(vk_x,gamma) e(C,delta) -- TxVerifier.sol:258-276 (pairingProd4 with -A)."""
    p = vk["nPublic"]
    if len(public_signals) != p:
        return False
    vk_x = vk["IC"][0]
    for i, x in enumerate(public_signals):
        if not 0 <= x < R:  # TxVerifier.sol:265 verifier-gte-snark-scalar-field
            return False
        vk_x = g1_add(vk_x, g1_mul(vk["IC"][i + 1], x))
    return pairing_product_is_one([
        (g1_neg(proof["pi_a"]), proof["pi_b"]),
        (vk["vk_alfa_1"], vk["vk_beta_2"]),
        (vk_x, vk["vk_gamma_2"]),
        (proof["pi_c"], vk["vk_delta_2"]),
    ])


# ----------------------------------------------------------------------------- codecs (binarify.ts)
def _le32(v):
    return int(v).to_bytes(32, "little")


def binarify_witness(witness):
    """binarify.ts:10-48 -- n x 32 B little-endian, standard (non-Montgomery) form."""
    return b"".join(_le32(x) for x in witness)


def _g1_json(P):
    return [0, 1, 0] if P is None else [P[0], P[1], 1]


def _g2_json(P):
    return [[0, 0], [1, 0], [0, 0]] if P is None else [[P[0][0], P[0][1]], [P[1][0], P[1][1]], [1, 0]]


def to_json_key(pk):
    """Render points in the snarkjs JSON shapes ([x,y,1]; infinity [0,1,0]); ints stay ints."""
    out = dict(pk)
    for f in ("A", "B1", "hExps"):
        out[f] = [_g1_json(P) for P in pk[f]]
    out["C"] = [None if i <= pk["nPublic"] else _g1_json(P) for i, P in enumerate(pk["C"])]
    out["B2"] = [_g2_json(P) for P in pk["B2"]]
    for f in ("vk_alfa_1", "vk_beta_1", "vk_delta_1"):
        out[f] = _g1_json(pk[f])
    for f in ("vk_beta_2", "vk_delta_2"):
        out[f] = _g2_json(pk[f])
    return out


def binarify_proving_key(jk):
    """binarify.ts:50-207 on a JSON-shaped key (to_json_key output).  All field elements Montgomery."""
    mq = lambda v: _le32(v * MONT % Q)
    mr = lambda v: _le32(v * MONT % R)
    u32 = lambda v: int(v).to_bytes(4, "little")
    pt = lambda P: mq(P[0]) + mq(P[1])
    pt2 = lambda P: mq(P[0][0]) + mq(P[0][1]) + mq(P[1][0]) + mq(P[1][1])

    def pol(d):
        keys = sorted(d.keys())  # JS Object.keys: integer-like keys ascend
        return u32(len(keys)) + b"".join(u32(k) + mr(d[k]) for k in keys)

    n, p, m = jk["nVars"], jk["nPublic"], jk["domainSize"]
    body = [pt(jk["vk_alfa_1"]), pt(jk["vk_beta_1"]), pt(jk["vk_delta_1"]), pt2(jk["vk_beta_2"]), pt2(jk["vk_delta_2"])]
    off = 40 + sum(len(x) for x in body)
    ptrs = []
    for sect in ([pol(jk["polsA"][i]) for i in range(n)], [pol(jk["polsB"][i]) for i in range(n)],
                 [pt(jk["A"][i]) for i in range(n)], [pt(jk["B1"][i]) for i in range(n)],
                 [pt2(jk["B2"][i]) for i in range(n)], [pt(jk["C"][i]) for i in range(p + 1, n)],
                 [pt(jk["hExps"][i]) for i in range(m)]):
        ptrs.append(off)
        blob = b"".join(sect)
        off += len(blob)
        body.append(blob)
    out = u32(n) + u32(p) + u32(m) + b"".join(u32(x) for x in ptrs) + b"".join(body)
    assert len(out) == off
    return out


def parse_proving_key(buf):
    """Inverse of binarify_proving_key -> dict with affine points / None and pols per signal."""
    u32 = lambda o: int.from_bytes(buf[o:o + 4], "little")
    rinv_q = inv(MONT % Q, Q)
    rinv_r = inv(MONT % R, R)
    fq = lambda o: int.from_bytes(buf[o:o + 32], "little") * rinv_q % Q
    n, p, m = u32(0), u32(4), u32(8)
    pA, pB, pPA, pPB1, pPB2, pPC, pPH = (u32(12 + 4 * i) for i in range(7))

    def g1(o):
        x, y = fq(o), fq(o + 32)
        return None if x == 0 and y == 1 else (x, y)

    def g2(o):
        x = (fq(o), fq(o + 32))
        y = (fq(o + 64), fq(o + 96))
        return None if x == (0, 0) and y == (1, 0) else (x, y)

    def pols(o):
        out = []
        for _ in range(n):
            k = u32(o)
            o += 4
            d = {}
            for _ in range(k):
                d[u32(o)] = int.from_bytes(buf[o + 4:o + 36], "little") * rinv_r % R
                o += 36
            out.append(d)
        return out, o

    polsA, endA = pols(pA)
    polsB, endB = pols(pB)
    assert endA == pB and endB == pPA
    pk = dict(nVars=n, nPublic=p, domainSize=m, polsA=polsA, polsB=polsB,
              vk_alfa_1=g1(40), vk_beta_1=g1(104), vk_delta_1=g1(168), vk_beta_2=g2(232), vk_delta_2=g2(360),
              A=[g1(pPA + 64 * i) for i in range(n)], B1=[g1(pPB1 + 64 * i) for i in range(n)],
              B2=[g2(pPB2 + 128 * i) for i in range(n)],
              C=[None] * (p + 1) + [g1(pPC + 64 * i) for i in range(n - p - 1)],
              hExps=[g1(pPH + 64 * i) for i in range(m)])
    assert pPH + 64 * m == len(buf)
    return pk


def proof_to_json(proof):
    """Shape returned by groth16GenProof: decimal strings."""
    a, b, c = proof["pi_a"], proof["pi_b"], proof["pi_c"]
    return dict(pi_a=[str(a[0]), str(a[1]), "1"],
                pi_b=[[str(b[0][0]), str(b[0][1])], [str(b[1][0]), str(b[1][1])], ["1", "0"]],
                pi_c=[str(c[0]), str(c[1]), "1"])


def solidity_proof(proof_json, public_signals):
    """operator/src/snarks/common.ts:43-50."""
    return dict(a=proof_json["pi_a"][:2], b=[list(reversed(x)) for x in proof_json["pi_b"]][:2],
                c=proof_json["pi_c"][:2], inputs=[str(x % R) for x in public_signals])


def proof_bytes(proof):
    """The C-ABI output layout (include/zkr.h): pi_a 64 B | pi_b 128 B | pi_c 64 B, LE standard form."""
    a, b, c = proof["pi_a"], proof["pi_b"], proof["pi_c"]
    return (_le32(a[0]) + _le32(a[1]) + _le32(b[0][0]) + _le32(b[0][1]) + _le32(b[1][0]) + _le32(b[1][1])
            + _le32(c[0]) + _le32(c[1]))


def sha256(b):
    return hashlib.sha256(b).hexdigest()


# ----------------------------------------------------------------------------- checker for device-generated keys
def parse_synth_aux(aux: bytes):
    """Decode the checker blob of zkr_synth_key (include/zkr.h): toxic waste, per-signal discrete logs
    a_s, b_s, c_s, the IC points and vk_gamma_2."""
    n = int.from_bytes(aux[:8], "little")
    o = 8
    rd = lambda k: int.from_bytes(aux[o + 32 * k:o + 32 * k + 32], "little")
    tox = dict(t=rd(0), alfa=rd(1), beta=rd(2), gamma=rd(3), delta=rd(4))
    o += 160
    vec = lambda: [int.from_bytes(aux[o + 32 * i:o + 32 * i + 32], "little") for i in range(n)]
    a = vec(); o += 32 * n
    b = vec(); o += 32 * n
    c = vec(); o += 32 * n
    rest = aux[o:]
    n_ic = (len(rest) - 128) // 64
    ic = [(int.from_bytes(rest[64 * i:64 * i + 32], "little"), int.from_bytes(rest[64 * i + 32:64 * i + 64], "little")) for i in range(n_ic)]
    g = rest[64 * n_ic:]
    gv = [int.from_bytes(g[32 * i:32 * i + 32], "little") for i in range(4)]
    return dict(n=n, tox=tox, a=a, b=b, c=c, ic=ic, gamma2=((gv[0], gv[1]), (gv[2], gv[3])))


def proof_from_aux(aux, witness_bytes, n_public, r, s, dot=None):
    """Toxic-waste closed form of the proof (satisfying witness) and the verifying key, for keys whose
    group elements were computed on the GPU: no MSM, no NTT -- field dot products + 3 scalar muls.
    dot(a_bytes, b_bytes) -> sum a_i b_i mod r over 32-byte LE arrays: pass coracle.fr_dot at sizes where a Python
    loop over big integers takes minutes (2^24: BASELINE configs[4]); default: the plain Python sums."""
    p = n_public
    if dot is None:
        ax = parse_synth_aux(aux)
        n, tox = ax["n"], ax["tox"]
        w = [int.from_bytes(witness_bytes[32 * i:32 * i + 32], "little") for i in range(n)]
        At = sum(x * y for x, y in zip(w, ax["a"])) % R
        Bt = sum(x * y for x, y in zip(w, ax["b"])) % R
        Ct = sum(x * y for x, y in zip(w, ax["c"])) % R
        head = (ax["a"][:p + 1], ax["b"][:p + 1], ax["c"][:p + 1])
        ic, gamma2 = ax["ic"], ax["gamma2"]
    else:  # the same sums over the byte arrays as they lie in the blob
        n = int.from_bytes(aux[:8], "little")
        rd = lambda o: int.from_bytes(aux[o:o + 32], "little")
        tox = dict(t=rd(8), alfa=rd(40), beta=rd(72), gamma=rd(104), delta=rd(136))
        o = 168
        wb = bytes(witness_bytes[:32 * n])
        sums, head = [], []
        for _ in range(3):
            sums.append(dot(wb, bytes(aux[o:o + 32 * n])))
            head.append([rd(o + 32 * i) for i in range(p + 1)])
            o += 32 * n
        At, Bt, Ct = sums
        w = [int.from_bytes(witness_bytes[32 * i:32 * i + 32], "little") for i in range(p + 1)]
        rest = aux[o:]
        n_ic = (len(rest) - 128) // 64
        ic = [(int.from_bytes(rest[64 * i:64 * i + 32], "little"), int.from_bytes(rest[64 * i + 32:64 * i + 64], "little")) for i in range(n_ic)]
        gv = [int.from_bytes(rest[64 * n_ic + 32 * i:64 * n_ic + 32 * i + 32], "little") for i in range(4)]
        gamma2 = ((gv[0], gv[1]), (gv[2], gv[3]))
    d = tox["delta"]
    dinv = inv(d, R)
    a_log = (tox["alfa"] + At + r * d) % R
    b_log = (tox["beta"] + Bt + s * d) % R
    pub = sum(w[i] * ((tox["beta"] * head[0][i] + tox["alfa"] * head[1][i] + head[2][i]) % R) for i in range(p + 1)) % R
    allk = (tox["beta"] * At + tox["alfa"] * Bt + Ct) % R
    cpriv = (allk - pub) * dinv % R
    c_log = (cpriv + (At * Bt - Ct) * dinv + s * a_log + r * b_log - r * s % R * d) % R
    fb1, fb2 = _fb()
    proof = dict(pi_a=fb1.mul(a_log), pi_b=fb2.mul(b_log), pi_c=fb1.mul(c_log))
    vk = dict(nPublic=p, IC=ic, vk_alfa_1=fb1.mul(tox["alfa"]), vk_beta_2=fb2.mul(tox["beta"]),
              vk_gamma_2=gamma2, vk_delta_2=fb2.mul(d))
    assert vk["vk_gamma_2"] == fb2.mul(tox["gamma"])
    return proof, vk, w[1:p + 1]
