// pairing.hpp -- BN254 optimal ate pairing on the HOST, for the Groth16 acceptance check.
//
// Replaces `groth.isValid(vk, proof, publicSignals)` (/root/reference/operator/src/snarks/common.ts:30-34, snarkjs)
// and mirrors what the on-chain verifier computes with precompile 8
// (/root/reference/contracts/contracts/TxVerifier.sol:91-115, 258-276).  SURVEY.md 8(a5) / 8(f-4): CPU only, a few
// milliseconds per proof, not a GPU target.  Built from the same field templates as the device code (field.hpp).
//
// Tower: Fq2 = Fq[u]/(u^2+1) (field.hpp), Fq6 = Fq2[v]/(v^3 - xi) with xi = 9 + u, Fq12 = Fq6[w]/(w^2 - v).
// D-type twist: (x', y') -> (x' w^2, y' w^3).  A line through the twisted point T with Fq2-slope lam, evaluated at
// P = (xP, yP) in G1, is  yP + (-lam xP) w + (lam xT - yT) w^3  =  ((yP, 0, 0), (-lam xP, lam xT - yT, 0)).
// Affine Miller loops over 6x+2 (all pairs in lockstep, one shared inversion per step) with the two Frobenius
// correction steps, final exponentiation by plain square-and-multiply -- the same mathematics as the checker in
// oracle/bn254.py, kept simple on purpose.
#pragma once
#include "curve.hpp"

#include <vector>
namespace zkr {
namespace pairing {

static inline Fq fq_from_limbs(const uint32_t (&l)[8]) {
  Fq r;
  for (int i = 0; i < 8; i++) r.v[i] = l[i];
  return to_mont(r);
}
static inline Fq fq_small(uint32_t x) {
  Fq r = Fq::zero();
  r.v[0] = x;
  return to_mont(r);
}
static inline Fq2 conj(const Fq2 &a) { return Fq2{a.a, neg(a.b)}; }
static inline Fq2 mul_fq(const Fq2 &a, const Fq &k) { return Fq2{mul(a.a, k), mul(a.b, k)}; }
static inline Fq2 mul_xi(const Fq2 &a) {  // (a0 + a1 u)(9 + u) = 9 a0 - a1 + (a0 + 9 a1) u
  Fq a8 = dbl(dbl(dbl(a.a))), b8 = dbl(dbl(dbl(a.b)));
  return Fq2{sub(add(a8, a.a), a.b), add(add(b8, a.b), a.a)};
}

struct Fq6 {
  Fq2 c0, c1, c2;
  static Fq6 zero() { return Fq6{Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
  static Fq6 one() { return Fq6{Fq2::one(), Fq2::zero(), Fq2::zero()}; }
  bool operator==(const Fq6 &o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
};
static inline Fq6 add(const Fq6 &a, const Fq6 &b) { return Fq6{add(a.c0, b.c0), add(a.c1, b.c1), add(a.c2, b.c2)}; }
static inline Fq6 sub(const Fq6 &a, const Fq6 &b) { return Fq6{sub(a.c0, b.c0), sub(a.c1, b.c1), sub(a.c2, b.c2)}; }
static inline Fq6 neg(const Fq6 &a) { return Fq6{neg(a.c0), neg(a.c1), neg(a.c2)}; }
static inline Fq6 mul(const Fq6 &a, const Fq6 &b) {
  Fq2 t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1), t2 = mul(a.c2, b.c2);
  Fq2 c0 = add(t0, mul_xi(sub(sub(mul(add(a.c1, a.c2), add(b.c1, b.c2)), t1), t2)));
  Fq2 c1 = add(sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), t0), t1), mul_xi(t2));
  Fq2 c2 = add(sub(sub(mul(add(a.c0, a.c2), add(b.c0, b.c2)), t0), t2), t1);
  return Fq6{c0, c1, c2};
}
static inline Fq6 mul_v(const Fq6 &a) { return Fq6{mul_xi(a.c2), a.c0, a.c1}; }  // (c0 + c1 v + c2 v^2) v
static inline Fq6 inv(const Fq6 &a) {
  Fq2 t0 = sub(sqr(a.c0), mul_xi(mul(a.c1, a.c2)));
  Fq2 t1 = sub(mul_xi(sqr(a.c2)), mul(a.c0, a.c1));
  Fq2 t2 = sub(sqr(a.c1), mul(a.c0, a.c2));
  Fq2 d = inv(add(mul(a.c0, t0), mul_xi(add(mul(a.c2, t1), mul(a.c1, t2)))));
  return Fq6{mul(t0, d), mul(t1, d), mul(t2, d)};
}

struct Fq12 {
  Fq6 c0, c1;  // c0 + c1 w
  static Fq12 one() { return Fq12{Fq6::one(), Fq6::zero()}; }
  bool operator==(const Fq12 &o) const { return c0 == o.c0 && c1 == o.c1; }
};
static inline Fq12 mul(const Fq12 &a, const Fq12 &b) {
  Fq6 t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1);
  return Fq12{add(t0, mul_v(t1)), sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), t0), t1)};
}
// (a0 + a1 w)^2 = a0^2 + v a1^2 + 2 a0 a1 w with two Fq6 products: (a0 + a1)(a0 + v a1) - a0 a1 - v a0 a1 is the first part
static inline Fq12 sqr(const Fq12 &a) {
  Fq6 ab = mul(a.c0, a.c1);
  Fq6 t = mul(add(a.c0, a.c1), add(a.c0, mul_v(a.c1)));
  return Fq12{sub(sub(t, ab), mul_v(ab)), add(ab, ab)};
}
// x * (b0 + b1 v): an Fq6 product whose second factor has no v^2 term (five Fq2 products instead of six)
static inline Fq6 mul_sparse01(const Fq6 &x, const Fq2 &b0, const Fq2 &b1) {
  Fq2 t0 = mul(x.c0, b0), t1 = mul(x.c1, b1);
  Fq2 c1 = sub(sub(mul(add(x.c0, x.c1), add(b0, b1)), t0), t1);
  return Fq6{add(t0, mul_xi(mul(x.c2, b1))), c1, add(t1, mul(x.c2, b0))};
}
// f * (a + (b0 + b1 v) w) for a in Fq: the line functions of the Miller loop (line() below builds exactly this shape): six Fq
// and ten Fq2 products instead of the eighteen Fq2 products of a general Fq12 product
static inline Fq12 mul_by_line(const Fq12 &f, const Fq &a, const Fq2 &b0, const Fq2 &b1) {
  Fq6 t0{mul_fq(f.c0.c0, a), mul_fq(f.c0.c1, a), mul_fq(f.c0.c2, a)};
  Fq6 t1 = mul_sparse01(f.c1, b0, b1);
  Fq6 m = mul_sparse01(add(f.c0, f.c1), Fq2{add(b0.a, a), b0.b}, b1);
  return Fq12{add(t0, mul_v(t1)), sub(sub(m, t0), t1)};
}
// a^2 for a in the cyclotomic subgroup (anything after the easy part of the final exponentiation): Granger-Scott, three
// Fq4 squarings = six Fq2 products instead of twelve
static inline Fq12 cyclotomic_sqr(const Fq12 &a) {
  Fq2 z0 = a.c0.c0, z4 = a.c0.c1, z3 = a.c0.c2, z2 = a.c1.c0, z1 = a.c1.c1, z5 = a.c1.c2;
  auto fq4_sqr = [](const Fq2 &x, const Fq2 &y, Fq2 &r0, Fq2 &r1) {  // (x + y s)^2 with s^2 = xi
    Fq2 tmp = mul(x, y);
    r0 = sub(sub(mul(add(x, y), add(x, mul_xi(y))), tmp), mul_xi(tmp));
    r1 = add(tmp, tmp);
  };
  Fq2 t0, t1, t2, t3, t4, t5;
  fq4_sqr(z0, z1, t0, t1);
  fq4_sqr(z2, z3, t2, t3);
  fq4_sqr(z4, z5, t4, t5);
  auto three_minus_two = [](const Fq2 &t, const Fq2 &z) { Fq2 d = sub(t, z); return add(add(d, d), t); };  // 3 t - 2 z
  auto three_plus_two = [](const Fq2 &t, const Fq2 &z) { Fq2 d = add(t, z); return add(add(d, d), t); };    // 3 t + 2 z
  z0 = three_minus_two(t0, z0);
  z1 = three_plus_two(t1, z1);
  z2 = three_plus_two(mul_xi(t5), z2);
  z3 = three_minus_two(t4, z3);
  z4 = three_minus_two(t2, z4);
  z5 = three_plus_two(t3, z5);
  return Fq12{Fq6{z0, z4, z3}, Fq6{z2, z1, z5}};
}
static inline Fq12 conj(const Fq12 &a) { return Fq12{a.c0, neg(a.c1)}; }
static inline Fq12 inv(const Fq12 &a) {
  Fq6 d = inv(sub(mul(a.c0, a.c0), mul_v(mul(a.c1, a.c1))));
  return Fq12{mul(a.c0, d), neg(mul(a.c1, d))};
}
template <int N>
static inline Fq12 pow(const Fq12 &a, const uint32_t (&e)[N]) {
  Fq12 r = Fq12::one();
  bool started = false;
  for (int i = 32 * N - 1; i >= 0; i--) {
    if (started) r = sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1) { r = started ? mul(r, a) : a; started = true; }
  }
  return r;
}

// the same for an element of the cyclotomic subgroup
template <int N>
static inline Fq12 pow_cyclotomic(const Fq12 &a, const uint32_t (&e)[N]) {
  Fq12 r = Fq12::one();
  bool started = false;
  for (int i = 32 * N - 1; i >= 0; i--) {
    if (started) r = cyclotomic_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1) { r = started ? mul(r, a) : a; started = true; }
  }
  return r;
}

// q^2 + 1 and (q^4 - q^2 + 1) / r, little-endian u32 limbs
static const uint32_t EXP_EASY2[16] = {0x275d69b2u, 0x3b5458a2u, 0x09eac101u, 0xa602072du, 0x6d96cadcu, 0x4a50189cu, 0x7a1242c8u, 0x04689e95u,
                                       0x34c6b38du, 0x26edfa5cu, 0x16375606u, 0xb00b8551u, 0x0348d21cu, 0x599a6f7cu, 0x763cbf9cu, 0x0925c4b8u};
static const uint32_t EXP_HARD[24] = {0xccdf42b1u, 0xe81bb482u, 0xf49c36d4u, 0x5abf5cc4u, 0x1da014fdu, 0xf1154e7eu, 0x87cdbacfu, 0xdcc7b44cu,
                                      0x954bcf8au, 0xaaa441e3u, 0xd5095f23u, 0x6b887d56u, 0xf3fd90c6u, 0x79581e16u, 0xd189227du, 0x3b1b1355u,
                                      0x61876f6bu, 0x4e529a58u, 0xd5b12278u, 0x6c0eb522u, 0x83177fafu, 0x331ec151u, 0x0b0759adu, 0x01baaa71u};
// 6x + 2, x = 4965661367192848881 (65 bits)
static const uint32_t ATE_LOOP[3] = {0xbe763ba8u, 0x9d797039u, 0x1u};
// xi^((q-1)/3), xi^((q-1)/2): Frobenius on the twist (standard form limbs, converted on first use)
static const uint32_t FROB_X0[8] = {0x176f553du, 0x99e39557u, 0xc2c3330cu, 0xb78cc310u, 0xf559b143u, 0x4c0bec3cu, 0x4f7911f7u, 0x2fb34798u};
static const uint32_t FROB_X1[8] = {0x640fcba2u, 0x1665d51cu, 0x0b7c9dceu, 0x32ae2a1du, 0xd75a0794u, 0x4ba4cc8bu, 0x61ebae20u, 0x16c9e550u};
static const uint32_t FROB_Y0[8] = {0x71a0135au, 0xdc540146u, 0xa9c95998u, 0xdbaae0edu, 0xb6e2f9b9u, 0xdc5ec698u, 0x489af5dcu, 0x063cf305u};
static const uint32_t FROB_Y1[8] = {0x2623b0e3u, 0x82d37f63u, 0x8fa25bd2u, 0x21807dc9u, 0xec796f2bu, 0x0704b5a7u, 0xac41049au, 0x07c03cbcu};
// twist constant b' = 3 / xi
static const uint32_t TWIST_B0[8] = {0x24a138e5u, 0x3267e6dcu, 0x59dbefa3u, 0xb5b4c5e5u, 0x1be06ac3u, 0x81be1899u, 0xceb8aaaeu, 0x2b149d40u};
static const uint32_t TWIST_B1[8] = {0x85c315d2u, 0xe4a2bd06u, 0xe52d1852u, 0xa74fa084u, 0xeed8fdf4u, 0xcd2cafadu, 0x3af0fed4u, 0x009713b0u};

static inline bool g1_on_curve(const G1Affine &p) {  // y^2 = x^3 + 3 (TxVerifier.sol:24-26 generator (1, 2))
  return sqr(p.y) == add(mul(sqr(p.x), p.x), fq_small(3));
}
static inline bool g2_on_curve(const G2Affine &p) {
  Fq2 b{fq_from_limbs(TWIST_B0), fq_from_limbs(TWIST_B1)};
  return sqr(p.y) == add(mul(sqr(p.x), p.x), b);
}

// Q is in the order-r subgroup G2 of the twist: [r]Q == infinity.  The twist's group order is r (2q - r) and the
// cofactor 2q - r has small prime factors (10069, ...), so an on-curve point need not be in G2; the bn256 pairing
// precompile the contracts call (TxVerifier.sol:91-115) rejects such points, and so must this verifier.
// g2_in_subgroup_plain is the definition (a 254-bit multiplication); g2_in_subgroup below is the test the verifier runs.
static inline bool g2_in_subgroup_plain(const G2Affine &p) {
  if (p.is_inf()) return true;
  G2XYZZ base = to_xyzz(p), acc = G2XYZZ::inf();
  for (int i = 253; i >= 0; i--) {  // r < 2^254
    acc = dbl_xyzz(acc);
    if ((FrParams::P[i >> 5] >> (i & 31)) & 1) acc = add_full(acc, base);
  }
  return acc.is_inf();
}

static inline G2Affine g2_frobenius(const G2Affine &p) {
  Fq2 gx{fq_from_limbs(FROB_X0), fq_from_limbs(FROB_X1)}, gy{fq_from_limbs(FROB_Y0), fq_from_limbs(FROB_Y1)};
  return G2Affine{mul(conj(p.x), gx), mul(conj(p.y), gy)};
}

// The same membership through the twist's Frobenius endomorphism psi (untwist, q-power, twist = g2_frobenius): on G2 psi acts
// as multiplication by q = 6x^2 + 1 - ... = 6x^2 (mod r) for the BN parameter x, and for a BN curve psi(Q) == [6x^2]Q holds
// ONLY on G2 (Scott, "A note on group membership tests for G1, G2 and GT on BLS pairing-friendly curves", and the BN case
// as implemented in gnark-crypto's bn254 G2 IsInSubGroup): a 127-bit multiplication instead of a 254-bit one.  Compared
// with the plain definition on subgroup points, on-curve points outside it and their sums in tests/test_host_arith.py.
static const uint32_t SIX_X_SQ[4] = {0xe87cfd46u, 0xf83e9682u, 0xeeb859fbu, 0x6f4d8248u};  // 6 x^2 = 147946756881789318990833708069417712966 = q - r
static inline bool g2_in_subgroup(const G2Affine &p) {
  if (p.is_inf()) return true;
  G2XYZZ base = to_xyzz(p), acc = G2XYZZ::inf();
  for (int i = 126; i >= 0; i--) {  // 6 x^2 < 2^127
    acc = dbl_xyzz(acc);
    if ((SIX_X_SQ[i >> 5] >> (i & 31)) & 1) acc = add_full(acc, base);
  }
  if (acc.is_inf()) return false;
  const G2Affine lhs = g2_frobenius(p);
  // [6x^2]Q == psi(Q) without leaving XYZZ: X == x ZZ and Y == y ZZZ
  return mul(lhs.x, acc.zz) == acc.x && mul(lhs.y, acc.zzz) == acc.y;
}

static inline Fq12 line(const G2Affine &t, const Fq2 &lam, const G1Affine &p) {
  Fq6 c0{Fq2{p.y, Fq::zero()}, Fq2::zero(), Fq2::zero()};
  Fq6 c1{mul_fq(lam, neg(p.x)), sub(mul(lam, t.x), t.y), Fq2::zero()};
  return Fq12{c0, c1};
}
// One step of n Miller loops in lockstep: the n slope denominators share ONE field inversion (Montgomery's trick),
// which is where an affine Miller loop spends most of its time.
constexpr int MAX_PAIRS = 8;
// false when a denominator is zero (the shared product would be zero and every inverse of the step with it)
static inline bool batch_inv(Fq2 *d, int n) {
  Fq2 pre[MAX_PAIRS];
  Fq2 run = Fq2::one();
  for (int i = 0; i < n; i++) {
    if (d[i].is_zero()) return false;
    pre[i] = run;
    run = mul(run, d[i]);
  }
  Fq2 iv = inv(run);
  for (int i = n - 1; i >= 0; i--) { Fq2 di = mul(iv, pre[i]); iv = mul(iv, d[i]); d[i] = di; }
  return true;
}
// doubling (s == nullptr) or addition of s[i]: t[i] <- step, f <- f * line_i for every live pair.  Returns false when
// a slope is undefined (T == +-S or 2T == infinity): that cannot happen for points of the order-r subgroup, which the
// callers check (g2_in_subgroup), so it is reported as "not a valid pairing input" instead of inverting zero.
static inline bool multi_step(G2Affine *t, const G2Affine *s, const G1Affine *p, const bool *live, int n, Fq12 &f) {
  Fq2 num[MAX_PAIRS], den[MAX_PAIRS];
  int idx[MAX_PAIRS], m = 0;
  for (int i = 0; i < n; i++) {
    if (!live[i]) continue;
    if (s) { num[m] = sub(s[i].y, t[i].y); den[m] = sub(s[i].x, t[i].x); }
    else { Fq2 xx = sqr(t[i].x); num[m] = add(dbl(xx), xx); den[m] = dbl(t[i].y); }
    idx[m++] = i;
  }
  if (!batch_inv(den, m)) return false;
  for (int k = 0; k < m; k++) {
    const int i = idx[k];
    Fq2 lam = mul(num[k], den[k]);
    f = mul_by_line(f, p[i].y, mul_fq(lam, neg(p[i].x)), sub(mul(lam, t[i].x), t[i].y));  // = f * line(t[i], lam, p[i])
    Fq2 x3 = sub(sub(sqr(lam), t[i].x), s ? s[i].x : t[i].x);
    Fq2 y3 = sub(mul(lam, sub(t[i].x, x3)), t[i].y);
    t[i] = G2Affine{x3, y3};
  }
  return true;
}

// prod_i f_{6x+2,Q_i}(P_i) with the two Frobenius correction lines; pairs with a point at infinity contribute 1.
// *ok (optional) is cleared when a step met an undefined slope (a Q outside the order-r subgroup): the value is
// then meaningless and the caller must treat the input as not valid.
static inline Fq12 multi_miller_loop(const G1Affine *ps, const G2Affine *qs, int n, bool *ok = nullptr) {
  G2Affine t[MAX_PAIRS], q1[MAX_PAIRS], q2[MAX_PAIRS];
  bool live[MAX_PAIRS];
  for (int i = 0; i < n; i++) {
    live[i] = !(qs[i].is_inf() || ps[i].is_inf());
    t[i] = qs[i];
    q1[i] = g2_frobenius(qs[i]);
    q2[i] = g2_frobenius(q1[i]);
    q2[i].y = neg(q2[i].y);
  }
  Fq12 f = Fq12::one();
  bool good = true;
  for (int b = 63; b >= 0 && good; b--) {  // bits below the leading one of the 65-bit loop count
    f = sqr(f);
    good = multi_step(t, nullptr, ps, live, n, f);
    if (good && ((ATE_LOOP[b >> 5] >> (b & 31)) & 1)) good = multi_step(t, qs, ps, live, n, f);
  }
  good = good && multi_step(t, q1, ps, live, n, f);
  good = good && multi_step(t, q2, ps, live, n, f);
  if (ok) *ok = good;
  return good ? f : Fq12::one();
}

// ---- The Miller loop the verifier runs (round 3): no inversions.
// The affine loop above pays one Fq2 inversion per step (~380 of a step's ~630 Fq products with four pairs).  Two kinds of pairs:
//  * FIXED second arguments (beta, gamma, delta of a verifying key): the slope and intercept of every step's line do not depend on
//    the first argument, so they are computed once per key (g2_prepare: the affine steps above, 89 lines) and a step costs two Fq
//    products for the evaluation at P and the sparse product with f;
//  * the proof's own B: homogeneous projective steps (Costello, Lange, Naehrig, "Faster pairing computations on curves with high-
//    degree twists"): a line may be scaled by any element of Fq2 -- it is killed by the final exponentiation, (q^2 - 1) divides
//    (q^12 - 1) / r -- so 2 Y Z, resp. the slope's denominator, multiplies the line instead of being inverted.
// With T = (X, Y, Z) on y^2 = x^3 + b':
//   doubling:  line = (2 Y Z) yP  -  (3 X^2) xP w  +  (Y^2 - 3 b' Z^2) w^3
//              X3 = (X Y / 2)(Y^2 - 9 b' Z^2),  Y3 = ((Y^2 + 9 b' Z^2) / 2)^2 - 27 b'^2 Z^4,  Z3 = 2 Y^3 Z
//   addition of the affine Q = (x2, y2), theta = Y - y2 Z, lam = X - x2 Z:
//              line = lam yP  -  theta xP w  +  (theta x2 - lam y2) w^3
//              X3 = lam H, Y3 = theta (X lam^2 - H) - lam^3 Y, Z3 = Z lam^3  with H = lam^3 + Z theta^2 - 2 X lam^2
// Same value after the final exponentiation as multi_miller_loop (tests/test_host_arith.py: random pairs, both loops).
struct G2Prepared {
  std::vector<Fq2> lam, c;  // per step of the loop over 6x + 2 and the two Frobenius steps: slope, slope * x_T - y_T
  bool inf = false;
};
static inline bool g2_prepare(const G2Affine &q, G2Prepared &out) {
  out.lam.clear(), out.c.clear();
  out.inf = q.is_inf();
  if (out.inf) return true;
  G2Affine t = q, q1 = g2_frobenius(q), q2 = g2_frobenius(q1);
  q2.y = neg(q2.y);
  auto step = [&](const G2Affine *s) -> bool {
    Fq2 num, den;
    if (s) { num = sub(s->y, t.y); den = sub(s->x, t.x); }
    else { Fq2 xx = sqr(t.x); num = add(dbl(xx), xx); den = dbl(t.y); }
    if (den.is_zero()) return false;
    Fq2 lam = mul(num, inv(den));
    out.lam.push_back(lam);
    out.c.push_back(sub(mul(lam, t.x), t.y));
    Fq2 x3 = sub(sub(sqr(lam), t.x), s ? s->x : t.x);
    t = G2Affine{x3, sub(mul(lam, sub(t.x, x3)), t.y)};
    return true;
  };
  for (int b = 63; b >= 0; b--) {
    if (!step(nullptr)) return false;
    if (((ATE_LOOP[b >> 5] >> (b & 31)) & 1) && !step(&q)) return false;
  }
  return step(&q1) && step(&q2);
}
// f * (a + (b0 + b1 v) w) with a in Fq2: thirteen Fq2 products
static inline Fq12 mul_by_line2(const Fq12 &f, const Fq2 &a, const Fq2 &b0, const Fq2 &b1) {
  Fq6 t0{mul(f.c0.c0, a), mul(f.c0.c1, a), mul(f.c0.c2, a)};
  Fq6 t1 = mul_sparse01(f.c1, b0, b1);
  Fq6 m = mul_sparse01(add(f.c0, f.c1), add(a, b0), b1);
  return Fq12{add(t0, mul_v(t1)), sub(sub(m, t0), t1)};
}
struct G2Proj { Fq2 x, y, z; };
static inline const Fq2 &twist_b() {
  static const Fq2 b{fq_from_limbs(TWIST_B0), fq_from_limbs(TWIST_B1)};
  return b;
}
static inline const Fq &fq_half() {
  static const Fq h = inv(fq_small(2));
  return h;
}
// T <- 2T; f <- f * line_{T,T}(P)
static inline void proj_double_step(G2Proj &t, const G1Affine &p, Fq12 &f) {
  const Fq2 xx = sqr(t.x), b = sqr(t.y), c = sqr(t.z);
  const Fq2 e = mul(twist_b(), add(dbl(c), c));                 // 3 b' Z^2
  const Fq2 f3 = add(dbl(e), e);                                // 9 b' Z^2
  const Fq2 a = mul_fq(mul(t.x, t.y), fq_half());               // X Y / 2
  const Fq2 g = mul_fq(add(b, f3), fq_half());                  // (Y^2 + 9 b' Z^2) / 2
  const Fq2 h = sub(sqr(add(t.y, t.z)), add(b, c));             // 2 Y Z
  const Fq2 e2 = sqr(e);
  f = mul_by_line2(f, mul_fq(h, p.y), mul_fq(add(dbl(xx), xx), neg(p.x)), sub(b, e));
  t = G2Proj{mul(a, sub(b, f3)), sub(sqr(g), add(dbl(e2), e2)), mul(b, h)};
}
// T <- T + Q (Q affine, T != +-Q); f <- f * line_{T,Q}(P)
static inline void proj_add_step(G2Proj &t, const G2Affine &q, const G1Affine &p, Fq12 &f) {
  const Fq2 theta = sub(t.y, mul(q.y, t.z)), lam = sub(t.x, mul(q.x, t.z));
  const Fq2 c = sqr(theta), d = sqr(lam), e = mul(lam, d), ff = mul(t.z, c), g = mul(t.x, d);
  const Fq2 h = sub(add(e, ff), dbl(g));
  f = mul_by_line2(f, mul_fq(lam, p.y), mul_fq(theta, neg(p.x)), sub(mul(theta, q.x), mul(lam, q.y)));
  t = G2Proj{mul(lam, h), sub(mul(theta, sub(g, h)), mul(e, t.y)), mul(t.z, e)};
}
// prod_i f_{6x+2,ql_i}(pl_i) * prod_j f_{6x+2,qf_j}(pf_j): nl pairs with arbitrary second arguments (projective steps), nf pairs
// with prepared ones.  Pairs with a point at infinity contribute 1.  *ok is cleared when a projective point degenerates (Z = 0:
// impossible for second arguments of order r, which the callers check).
static inline Fq12 miller_loop_mixed(const G1Affine *pl, const G2Affine *ql, size_t nl, const G1Affine *pf, const G2Prepared *const *qf, size_t nf, bool *ok = nullptr) {
  std::vector<G2Proj> t(nl);
  std::vector<G2Affine> q1(nl), q2(nl);
  std::vector<char> live(nl), livef(nf);
  for (size_t i = 0; i < nl; i++) {
    live[i] = !(ql[i].is_inf() || pl[i].is_inf());
    t[i] = G2Proj{ql[i].x, ql[i].y, Fq2::one()};
    q1[i] = g2_frobenius(ql[i]);
    q2[i] = g2_frobenius(q1[i]);
    q2[i].y = neg(q2[i].y);
  }
  for (size_t j = 0; j < nf; j++) livef[j] = !(qf[j]->inf || pf[j].is_inf());
  Fq12 f = Fq12::one();
  size_t at = 0;  // index of the step in the prepared tables
  auto fixed_lines = [&]() {
    for (size_t j = 0; j < nf; j++)
      if (livef[j]) f = mul_by_line(f, pf[j].y, mul_fq(qf[j]->lam[at], neg(pf[j].x)), qf[j]->c[at]);
    at++;
  };
  for (int b = 63; b >= 0; b--) {
    f = sqr(f);
    for (size_t i = 0; i < nl; i++)
      if (live[i]) proj_double_step(t[i], pl[i], f);
    fixed_lines();
    if ((ATE_LOOP[b >> 5] >> (b & 31)) & 1) {
      for (size_t i = 0; i < nl; i++)
        if (live[i]) proj_add_step(t[i], ql[i], pl[i], f);
      fixed_lines();
    }
  }
  for (size_t i = 0; i < nl; i++)
    if (live[i]) proj_add_step(t[i], q1[i], pl[i], f);
  fixed_lines();
  for (size_t i = 0; i < nl; i++)
    if (live[i]) proj_add_step(t[i], q2[i], pl[i], f);
  fixed_lines();
  bool good = true;
  for (size_t i = 0; i < nl; i++)
    if (live[i] && t[i].z.is_zero()) good = false;  // the loop ends at -pi^3(Q), never at infinity, for Q of order r
  if (ok) *ok = good;
  return f;
}

// f^((q^12 - 1) / r) by plain square-and-multiply over the 1270 exponent bits: the reference form the fast one below is
// tested against (tests/test_host_arith.py through the host shim)
static inline Fq12 final_exponentiation_plain(const Fq12 &f0) {
  Fq12 f = mul(conj(f0), inv(f0));  // ^(q^6 - 1)
  f = pow(f, EXP_EASY2);            // ^(q^2 + 1)
  return pow(f, EXP_HARD);          // ^((q^4 - q^2 + 1) / r)
}

// Frobenius maps on Fq12 = Fq2[w]/(w^6 - xi): (sum a_i w^i)^q = sum conj(a_i) g1[i] w^i with g1[i] = xi^(i (q-1)/6),
// (..)^(q^2) = sum a_i g2[i] w^i with g2[i] = g1[i] conj(g1[i]) in Fq.  Constants derived once from xi.
struct FrobConst {
  Fq2 g1[6];
  Fq g2[6];
  FrobConst() {
    // e = (q - 1) / 6 by long division of the limbs
    uint32_t e[8];
    uint64_t rem = 0;
    uint32_t qm1[8];
    for (int i = 0; i < 8; i++) qm1[i] = FqParams::P[i];
    qm1[0] -= 1;
    for (int i = 7; i >= 0; i--) {
      uint64_t cur = (rem << 32) | qm1[i];
      e[i] = (uint32_t)(cur / 6);
      rem = cur % 6;
    }
    Fq2 xi{fq_small(9), fq_small(1)}, r = Fq2::one(), b = xi;
    for (int i = 0; i < 256; i++) {
      if ((e[i >> 5] >> (i & 31)) & 1) r = mul(r, b);
      b = sqr(b);
    }
    g1[0] = Fq2::one();
    for (int i = 1; i < 6; i++) g1[i] = mul(g1[i - 1], r);
    for (int i = 0; i < 6; i++) g2[i] = sub(mul(g1[i].a, g1[i].a), neg(mul(g1[i].b, g1[i].b)));  // norm a^2 + b^2
  }
};
static inline const FrobConst &frob_const() {
  static const FrobConst k;
  return k;
}
// coefficient of w^i: i = 0, 2, 4 -> c0.{c0, c1, c2}; i = 1, 3, 5 -> c1.{c0, c1, c2}   (v = w^2)
static inline Fq12 frobenius(const Fq12 &a) {
  const FrobConst &k = frob_const();
  return Fq12{Fq6{conj(a.c0.c0), mul(conj(a.c0.c1), k.g1[2]), mul(conj(a.c0.c2), k.g1[4])},
              Fq6{mul(conj(a.c1.c0), k.g1[1]), mul(conj(a.c1.c1), k.g1[3]), mul(conj(a.c1.c2), k.g1[5])}};
}
static inline Fq12 frobenius2(const Fq12 &a) {
  const FrobConst &k = frob_const();
  return Fq12{Fq6{a.c0.c0, mul_fq(a.c0.c1, k.g2[2]), mul_fq(a.c0.c2, k.g2[4])},
              Fq6{mul_fq(a.c1.c0, k.g2[1]), mul_fq(a.c1.c1, k.g2[3]), mul_fq(a.c1.c2, k.g2[5])}};
}
// the curve parameter x = 4965661367192848881 (ATE_LOOP = 6x + 2)
static const uint32_t BN_X[2] = {0x4a6909f1u, 0x44e992b4u};

// The same value with the structure of the exponent: q^6 - 1 by conjugation and one inversion, q^2 + 1 by a Frobenius
// map, the hard part (q^4 - q^2 + 1)/r as lambda_3 q^3 + lambda_2 q^2 + lambda_1 q + lambda_0 with the lambda_i
// polynomials in x (Scott, Benger, Charlemagne, Dominguez Perez, Kachisa: three powers by x, Frobenius maps and the
// addition chain y0 y1^2 y2^6 y3^12 y4^18 y5^30 y6^36; inverses are conjugates after the easy part).
static inline Fq12 final_exponentiation(const Fq12 &f0) {
  Fq12 f = mul(conj(f0), inv(f0));  // ^(q^6 - 1)
  f = mul(frobenius2(f), f);        // ^(q^2 + 1)
  Fq12 fx = pow_cyclotomic(f, BN_X), fx2 = pow_cyclotomic(fx, BN_X), fx3 = pow_cyclotomic(fx2, BN_X);
  Fq12 fp = frobenius(f), fp2 = frobenius2(f), fp3 = frobenius(fp2);
  Fq12 y0 = mul(mul(fp, fp2), fp3);
  Fq12 y1 = conj(f);
  Fq12 y2 = frobenius2(fx2);
  Fq12 y3 = conj(frobenius(fx));
  Fq12 y4 = conj(mul(fx, frobenius(fx2)));
  Fq12 y5 = conj(fx2);
  Fq12 y6 = conj(mul(fx3, frobenius(fx3)));
  Fq12 t0 = mul(mul(cyclotomic_sqr(y6), y4), y5);
  Fq12 t1 = mul(mul(y3, y5), t0);
  t0 = mul(t0, y2);
  t1 = cyclotomic_sqr(mul(cyclotomic_sqr(t1), t0));
  t0 = mul(t1, y1);
  t1 = mul(t1, y0);
  return mul(cyclotomic_sqr(t0), t1);
}

// prod_i e(P_i, Q_i) == 1  -- the bn256 pairing precompile's check (TxVerifier.sol:91-115)
static inline bool pairing_product_is_one(const G1Affine *ps, const G2Affine *qs, int n) {
  if (n > MAX_PAIRS) return false;
  bool ok = true;
  Fq12 f = multi_miller_loop(ps, qs, n, &ok);
  return ok && final_exponentiation(f) == Fq12::one();
}

}  // namespace pairing
}  // namespace zkr
