// host_arith_shim.cpp -- exposes the device field/curve templates, compiled for the HOST, through a
// tiny C ABI so tests/test_host_arith.py can check them against the oracle without a GPU.
// Not part of the product path (the product is libzkr_hip.so).
#include "hostops.hpp"
#include "pairing.hpp"
using namespace zkr;

extern "C" {
// op: 0 mul, 1 add, 2 sub, 3 inv(a), 4 neg(a), 5 sqr(a); inputs/outputs standard form
void zkt_fp(int field, int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
  if (field == 0) {
    Fq x = to_mont(load_fp<FqParams>(a)), y = to_mont(load_fp<FqParams>(b)), r;
    switch (op) { case 0: r = mul(x, y); break; case 1: r = add(x, y); break; case 2: r = sub(x, y); break;
                  case 3: r = inv(x); break; case 4: r = neg(x); break; default: r = sqr(x); }
    store_fp(out, from_mont(r));
  } else {
    Fr x = to_mont(load_fp<FrParams>(a)), y = to_mont(load_fp<FrParams>(b)), r;
    switch (op) { case 0: r = mul(x, y); break; case 1: r = add(x, y); break; case 2: r = sub(x, y); break;
                  case 3: r = inv(x); break; case 4: r = neg(x); break; default: r = sqr(x); }
    store_fp(out, from_mont(r));
  }
}
// a, b: 64 B standard (re, im); op 0 mul, 1 sqr(a), 2 inv(a)
void zkt_fq2(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
  Fq2 x{to_mont(load_fp<FqParams>(a)), to_mont(load_fp<FqParams>(a + 32))};
  Fq2 y{to_mont(load_fp<FqParams>(b)), to_mont(load_fp<FqParams>(b + 32))};
  Fq2 r = op == 0 ? mul(x, y) : op == 1 ? sqr(x) : inv(x);
  store_fp(out, from_mont(r.a));
  store_fp(out + 32, from_mont(r.b));
}
// k*P with mixed adds only (exercises add_mixed incl. its doubling branch) -> std affine; returns 1 if infinity
int zkt_g1_mul(const uint8_t *p_mont, const uint8_t *k32, uint8_t *out) {
  G1Affine p = load_g1(p_mont);
  U256 k = load_u256(k32);
  G1XYZZ acc = G1XYZZ::inf();
  for (int i = 255; i >= 0; i--) {
    acc = dbl_xyzz(acc);
    if ((k.v[i >> 5] >> (i & 31)) & 1) acc = add_mixed(acc, p);
  }
  if (acc.is_inf()) return 1;
  store_g1_std(out, to_affine(acc));
  return 0;
}
int zkt_g2_mul(const uint8_t *p_mont, const uint8_t *k32, uint8_t *out) {
  G2Affine p = load_g2(p_mont);
  U256 k = load_u256(k32);
  G2XYZZ acc = scalar_mul(to_xyzz(p), k);
  if (acc.is_inf()) return 1;
  store_g2_std(out, to_affine(acc));
  return 0;
}
// (a*P) + (b*P) through add_full / add_mixed(neg) corner cases: returns std affine of a*P + sign*b*P
int zkt_g1_lincomb(const uint8_t *p_mont, uint32_t a, uint32_t b, int negate_b, uint8_t *out) {
  G1Affine p = load_g1(p_mont);
  G1XYZZ pa = mul_small(to_xyzz(p), a), pb = mul_small(to_xyzz(p), b);
  G1XYZZ r;
  if (!pb.is_inf()) {
    G1Affine qb = to_affine(pb);
    r = add_mixed(pa, qb, negate_b != 0);
  } else {
    r = pa;
  }
  if (r.is_inf()) return 1;
  store_g1_std(out, to_affine(r));
  return 0;
}
// The verifier's structured final exponentiation (Frobenius maps + three powers by x) against plain square-and-multiply
// over (q^12 - 1)/r, and the Frobenius maps against powers by q, on `count` pseudo-random Fq12 elements (xorshift from
// `seed`).  Returns the number of elements on which everything agrees.
int zkt_final_exp_check(uint64_t seed, int count) {
  using namespace zkr::pairing;
  uint64_t st = seed ? seed : 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
  auto rfq = [&]() { Fq a; for (int i = 0; i < 8; i++) a.v[i] = rnd(); a.v[7] &= 0x0fffffffu; return to_mont(a); };
  auto rfq2 = [&]() { Fq2 r; r.a = rfq(); r.b = rfq(); return r; };
  auto rfq6 = [&]() { Fq6 r; r.c0 = rfq2(); r.c1 = rfq2(); r.c2 = rfq2(); return r; };
  uint32_t q[8];
  for (int i = 0; i < 8; i++) q[i] = FqParams::P[i];
  int ok = 0;
  for (int it = 0; it < count; it++) {
    Fq12 f{rfq6(), rfq6()};
    Fq12 fq = pow(f, q);
    ok += (fq == frobenius(f)) && (pow(fq, q) == frobenius2(f)) && (final_exponentiation_plain(f) == final_exponentiation(f));
  }
  return ok;
}
// proof-assembly helpers (hostops.hpp) against plain double-and-add: fixed-base window table, joint double
// multiplication; scalars a, b standard 32 B.  Returns 1 when both agree.
int zkt_assembly_muls(const uint8_t *p_mont, const uint8_t *q_mont, const uint8_t *a32, const uint8_t *b32) {
  G1Affine p = load_g1(p_mont), q = load_g1(q_mont);
  U256 a = load_u256(a32), b = load_u256(b32);
  G1XYZZ want_fixed = scalar_mul(to_xyzz(p), a);
  G1XYZZ got_fixed = fixed_base_mul(fixed_base_table(p), a);
  G1XYZZ want_joint = add_full(scalar_mul(to_xyzz(p), a), scalar_mul(to_xyzz(q), b));
  G1XYZZ got_joint = double_scalar_mul(to_xyzz(p), a, to_xyzz(q), b);
  auto same = [](const G1XYZZ &x, const G1XYZZ &y) {
    if (x.is_inf() || y.is_inf()) return x.is_inf() == y.is_inf();
    G1Affine u = to_affine(x), v = to_affine(y);
    return u.x == v.x && u.y == v.y;
  };
  return same(want_fixed, got_fixed) && same(want_joint, got_joint);
}
}
