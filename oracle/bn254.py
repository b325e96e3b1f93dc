"""BN254 (alt_bn128) arithmetic -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The product path is the HIP library
(simple-zk-rollups_amd/csrc) and fails loudly when that library is missing.

PARITY UNPINNED (proof bytes): the reference (kendricktan/simple-zk-rollups) holds no
golden proof/NTT/MSM vector -- every test draws random inputs and blinding
(SURVEY.md 8(c)).  What IS pinned against the reference tree:
  * moduli q, r                    operator/src/utils/binarify.ts:79-81,86-88,
                                   contracts/contracts/TxVerifier.sol:50,259
  * G1 generator (1,2), G2 generator contracts/contracts/TxVerifier.sol:24-35
    (Solidity limb order is [im, re]; snarkjs order is [re, im])
  * the 80 G1 / 6 G2 verifying-key constants of TxVerifier.sol:176-257 and
    WithdrawVerifier.sol (tests/golden/verifier_points.json) must decode on-curve
    and in the r-torsion under this file's curve/twist equations
  * the verification equation     contracts/contracts/TxVerifier.sol:258-276
The arithmetic itself lives in un-vendored npm dependencies (snarkjs 0.1.20,
websnark 0.0.5); this file restates their published algorithms.

Plain Python ints.  Points: G1 affine = (x, y) or None (infinity);
G2 affine = ((x0, x1), (y0, y1)) with Fq2 element a0 + a1*u, u^2 = -1, or None.
"""

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BN_X = 4965661367192848881  # curve parameter; q = 36x^4+36x^3+24x^2+6x+1
ATE_LOOP = 6 * BN_X + 2

G1_GEN = (1, 2)
# snarkjs order [re, im]; TxVerifier.sol:30-35 lists [im, re]
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)

B1 = 3


def inv(a, p=Q):
    return pow(a, p - 2, p)


# ----------------------------------------------------------------------------- Fq2
def f2add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


def f2sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def f2neg(a):
    return (-a[0] % Q, -a[1] % Q)


def f2mul(a, b):
    t0 = a[0] * b[0]
    t1 = a[1] * b[1]
    return ((t0 - t1) % Q, ((a[0] + a[1]) * (b[0] + b[1]) - t0 - t1) % Q)


def f2sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % Q, 2 * a[0] * a[1] % Q)


def f2scal(a, k):
    return (a[0] * k % Q, a[1] * k % Q)


def f2inv(a):
    d = inv((a[0] * a[0] + a[1] * a[1]) % Q)
    return (a[0] * d % Q, -a[1] * d % Q)


def f2conj(a):
    return (a[0], -a[1] % Q)


def f2pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2mul(r, a)
        a = f2sqr(a)
        e >>= 1
    return r


F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (9, 1)  # non-residue for the sextic twist
B2 = f2mul((3, 0), f2inv(XI))  # twist coefficient b' = 3/(9+u)


# ----------------------------------------------------------------------------- G1
def g1_is_on_curve(P):
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - B1) % Q == 0


def g1_neg(P):
    return None if P is None else (P[0], -P[1] % Q)


def g1_add(P, S):
    if P is None:
        return S
    if S is None:
        return P
    x1, y1 = P
    x2, y2 = S
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * inv(2 * y1) % Q
    else:
        lam = (y2 - y1) * inv((x2 - x1) % Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return (x3, (lam * (x1 - x3) - y1) % Q)


# Jacobian internals for speed in scalar multiplication
def _jdbl(X, Y, Z):
    if Y == 0 or Z == 0:
        return (1, 1, 0)
    A = X * X % Q
    B = Y * Y % Q
    C = B * B % Q
    D = 2 * ((X + B) * (X + B) - A - C) % Q
    E = 3 * A % Q
    X3 = (E * E - 2 * D) % Q
    Y3 = (E * (D - X3) - 8 * C) % Q
    Z3 = 2 * Y * Z % Q
    return (X3, Y3, Z3)


def _jadd_mixed(X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % Q
    U2 = x2 * Z1Z1 % Q
    S2 = y2 * Z1 * Z1Z1 % Q
    H = (U2 - X1) % Q
    rr = (S2 - Y1) % Q
    if H == 0:
        if rr == 0:
            return _jdbl(X1, Y1, Z1)
        return (1, 1, 0)
    HH = H * H % Q
    HHH = H * HH % Q
    V = X1 * HH % Q
    X3 = (rr * rr - HHH - 2 * V) % Q
    Y3 = (rr * (V - X3) - Y1 * HHH) % Q
    Z3 = Z1 * H % Q
    return (X3, Y3, Z3)


def _jaffine(X, Y, Z):
    if Z == 0:
        return None
    zi = inv(Z)
    zi2 = zi * zi % Q
    return (X * zi2 % Q, Y * zi2 * zi % Q)


def g1_mul(P, k):
    """k*P, k reduced mod r (all points used here lie in the order-r group)."""
    k %= R
    if P is None or k == 0:
        return None
    x, y = P
    acc = (1, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jdbl(*acc)
        if bit == "1":
            acc = _jadd_mixed(*acc, x, y)
    return _jaffine(*acc)


def g1_msm(points, scalars):
    """Naive sum_i scalars[i]*points[i] (the snarkjs 0.1.20 genProof loop)."""
    acc = None
    for P, k in zip(points, scalars):
        if P is None or k % R == 0:
            continue
        acc = g1_add(acc, g1_mul(P, k))
    return acc


# ----------------------------------------------------------------------------- G2
def g2_is_on_curve(P):
    if P is None:
        return True
    x, y = P
    return f2sub(f2sqr(y), f2add(f2mul(f2sqr(x), x), B2)) == F2_ZERO


def g2_neg(P):
    return None if P is None else (P[0], f2neg(P[1]))


def g2_add(P, S):
    if P is None:
        return S
    if S is None:
        return P
    x1, y1 = P
    x2, y2 = S
    if x1 == x2:
        if f2add(y1, y2) == F2_ZERO:
            return None
        lam = f2mul(f2scal(f2sqr(x1), 3), f2inv(f2scal(y1, 2)))
    else:
        lam = f2mul(f2sub(y2, y1), f2inv(f2sub(x2, x1)))
    x3 = f2sub(f2sub(f2sqr(lam), x1), x2)
    return (x3, f2sub(f2mul(lam, f2sub(x1, x3)), y1))


def g2_mul(P, k, reduce=True):
    if reduce:
        k %= R
    if P is None or k == 0:
        return None
    acc = None
    for bit in bin(k)[2:]:
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, P)
    return acc


def g2_msm(points, scalars):
    acc = None
    for P, k in zip(points, scalars):
        if P is None or k % R == 0:
            continue
        acc = g2_add(acc, g2_mul(P, k))
    return acc


# ----------------------------------------------------------------------------- Fq6 / Fq12 tower
# Fq6 = Fq2[v]/(v^3 - XI), element (c0, c1, c2); Fq12 = Fq6[w]/(w^2 - v), element (a, b) = a + b*w
def _mulxi(a):
    # (a0 + a1 u)(9 + u) = 9a0 - a1 + (a0 + 9a1) u
    return ((9 * a[0] - a[1]) % Q, (a[0] + 9 * a[1]) % Q)


F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)


def f6add(a, b):
    return (f2add(a[0], b[0]), f2add(a[1], b[1]), f2add(a[2], b[2]))


def f6sub(a, b):
    return (f2sub(a[0], b[0]), f2sub(a[1], b[1]), f2sub(a[2], b[2]))


def f6neg(a):
    return (f2neg(a[0]), f2neg(a[1]), f2neg(a[2]))


def f6mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0 = f2mul(a0, b0)
    t1 = f2mul(a1, b1)
    t2 = f2mul(a2, b2)
    c0 = f2add(t0, _mulxi(f2sub(f2sub(f2mul(f2add(a1, a2), f2add(b1, b2)), t1), t2)))
    c1 = f2add(f2sub(f2sub(f2mul(f2add(a0, a1), f2add(b0, b1)), t0), t1), _mulxi(t2))
    c2 = f2add(f2sub(f2sub(f2mul(f2add(a0, a2), f2add(b0, b2)), t0), t2), t1)
    return (c0, c1, c2)


def f6mulv(a):
    # multiply by v: (c0 + c1 v + c2 v^2) v = xi*c2 + c0 v + c1 v^2
    return (_mulxi(a[2]), a[0], a[1])


def f6inv(a):
    a0, a1, a2 = a
    t0 = f2sub(f2sqr(a0), _mulxi(f2mul(a1, a2)))
    t1 = f2sub(_mulxi(f2sqr(a2)), f2mul(a0, a1))
    t2 = f2sub(f2sqr(a1), f2mul(a0, a2))
    d = f2inv(f2add(f2mul(a0, t0), _mulxi(f2add(f2mul(a2, t1), f2mul(a1, t2)))))
    return (f2mul(t0, d), f2mul(t1, d), f2mul(t2, d))


F12_ONE = (F6_ONE, F6_ZERO)


def f12mul(a, b):
    a0, a1 = a
    b0, b1 = b
    t0 = f6mul(a0, b0)
    t1 = f6mul(a1, b1)
    c0 = f6add(t0, f6mulv(t1))
    c1 = f6sub(f6sub(f6mul(f6add(a0, a1), f6add(b0, b1)), t0), t1)
    return (c0, c1)


def f12sqr(a):
    return f12mul(a, a)


def f12conj(a):
    return (a[0], f6neg(a[1]))


def f12inv(a):
    a0, a1 = a
    d = f6inv(f6sub(f6mul(a0, a0), f6mulv(f6mul(a1, a1))))
    return (f6mul(a0, d), f6neg(f6mul(a1, d)))


def f12pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12sqr(r)
        if bit == "1":
            r = f12mul(r, a)
    return r


# ----------------------------------------------------------------------------- optimal ate pairing
# Untwist (D-type): (x', y') -> (x' w^2, y' w^3), w^6 = XI.  A line through twisted T with
# Fq2-slope lam evaluated at P=(xP,yP) in G1 is  yP + (-lam*xP) w + (lam*xT - yT) w^3,
# i.e. Fq12 element ((yP,0,0), (-lam*xP, lam*xT - yT, 0)) in the tower above (w^3 = v*w).
_FROB_X = f2pow(XI, (Q - 1) // 3)  # gamma_{1,2}
_FROB_Y = f2pow(XI, (Q - 1) // 2)  # gamma_{1,3}


def _g2_frobenius(P):
    return (f2mul(f2conj(P[0]), _FROB_X), f2mul(f2conj(P[1]), _FROB_Y))


def _line(T, lam, P):
    xP, yP = P
    c1 = (f2scal(lam, -xP % Q), f2sub(f2mul(lam, T[0]), T[1]), F2_ZERO)
    return (((yP, 0), F2_ZERO, F2_ZERO), c1)


def _dbl_step(T, P):
    lam = f2mul(f2scal(f2sqr(T[0]), 3), f2inv(f2scal(T[1], 2)))
    l = _line(T, lam, P)
    x3 = f2sub(f2sqr(lam), f2scal(T[0], 2))
    y3 = f2sub(f2mul(lam, f2sub(T[0], x3)), T[1])
    return (x3, y3), l


def _add_step(T, S, P):
    lam = f2mul(f2sub(S[1], T[1]), f2inv(f2sub(S[0], T[0])))
    l = _line(T, lam, P)
    x3 = f2sub(f2sub(f2sqr(lam), T[0]), S[0])
    y3 = f2sub(f2mul(lam, f2sub(T[0], x3)), T[1])
    return (x3, y3), l


def miller_loop(Qp, P):
    """Miller function f_{6x+2,Q}(P) with the two Frobenius correction lines (G2 point Qp, G1 point P)."""
    if Qp is None or P is None:
        return F12_ONE
    T = Qp
    f = F12_ONE
    for bit in bin(ATE_LOOP)[3:]:
        T, l = _dbl_step(T, P)
        f = f12mul(f12sqr(f), l)
        if bit == "1":
            T, l = _add_step(T, Qp, P)
            f = f12mul(f, l)
    Q1 = _g2_frobenius(Qp)
    Q2 = g2_neg(_g2_frobenius(Q1))
    T, l = _add_step(T, Q1, P)
    f = f12mul(f, l)
    T, l = _add_step(T, Q2, P)
    f = f12mul(f, l)
    return f


def final_exponentiation(f):
    # easy part: f^(q^6-1) = conj(f)/f, then ^(q^2+1); hard part ^((q^4-q^2+1)/r)
    f = f12mul(f12conj(f), f12inv(f))
    f = f12pow(f, Q * Q + 1)
    return f12pow(f, (Q ** 4 - Q * Q + 1) // R)


def pairing(Qp, P):
    return final_exponentiation(miller_loop(Qp, P))


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 for pairs [(P in G1, Q in G2)] -- the bn256 precompile-8 check
    used by contracts/contracts/TxVerifier.sol:91-115."""
    f = F12_ONE
    for P, Qp in pairs:
        f = f12mul(f, miller_loop(Qp, P))
    return final_exponentiation(f) == F12_ONE
