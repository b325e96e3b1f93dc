// host_arith_shim.cpp -- exposes the device field/curve templates, compiled for the HOST, through a
// tiny C ABI so tests/test_host_arith.py can check them against the oracle without a GPU.
// Not part of the product path (the product is libzkr_hip.so).
#include "hostops.hpp"
#include "pairing.hpp"
#include "curve29.hpp"
using namespace zkr;

// ---- the 29-bit-limb hot-path arithmetic (field29.hpp, curve29.hpp) compiled for the host
namespace {
template <class P29> L29<P29, 3> to_m261(const uint8_t *std32) {  // standard form -> x 2^261
  uint32_t w[8];
  memcpy(w, std32, 32);
  return mul(unpack29<P29, 10>(w), const29<P29>(P29::R2)).template to<3>();
}
template <class P29, int H> void from_m261(const L29<P29, H> &x, uint8_t *std32) {  // x 2^261 (any bound a product accepts) -> canonical standard form
  L29<P29, 2> raw_one = L29<P29, 2>::zero();
  raw_one.v[0] = 1;
  uint32_t w[8];
  pack29(canonical(mul(x, raw_one)), w);
  memcpy(std32, w, 32);
}
template <class P29> void f29_op(int op, const uint8_t *a32, const uint8_t *b32, uint8_t *out) {
  auto a = to_m261<P29>(a32), b = to_m261<P29>(b32);
  switch (op) {
    case 0: from_m261(mul(a, b), out); break;
    case 1: from_m261(sqr(a), out); break;
    case 2: from_m261(add(a, b), out); break;
    case 3: from_m261(sub(a, b), out); break;
    case 4: from_m261(neg(a), out); break;
    case 5: from_m261(mul_sub(a, b, b, a), out); break;                                  // 0
    case 6: from_m261(mul_sum2(a, a, b, b), out); break;                                 // a^2 + b^2
    case 7: {  // values at the wide end of their bounds: ((a - b) - (b - a) + 2 (a + b))^2 with lazy intermediates
      auto t = sub(sub(a, b), sub(b, a));                 // 2 (a - b), bound 3 + 4 + 8 = 15 half moduli
      auto u = add(t, dbl(add(a, b)));                    // + 2 (a + b): 4 a, bound 15 + 12 = 27... too wide for a square: reduce
      from_m261(sqr(weak(u)), out);                       // 16 a^2
      break;
    }
    case 8: from_m261(mul_sum4(a, b, a, a, b, b, neg(a), b), out); break;               // a b + a^2 + b^2 - a b
    case 9: from_m261(weak(sub(sub(sub(a, b), b), b)), out); break;                      // a - 3 b through three lazy subtractions
    case 11: from_m261(mul(a, neg_loose(b)), out); break;                                 // -a b, the negation without its carry sweep
    case 12: from_m261(mul(b, sub_loose(a, b)), out); break;                              // b (a - b)
    case 13: from_m261(mul_sum2(a, sub_loose(a, b), neg_loose(b), a), out); break;        // a (a - b) - a b: the shape of the Y coordinate
    case 14: from_m261(mul_cneg(a, true, b), out); break;                                 // -a b through the per-lane sign select
    case 15: from_m261(mul_cneg(a, false, b), out); break;                                // a b
    default: from_m261(mul(sub(a, b).template to<17>(), sub(b, a).template to<17>()), out);  // -(a - b)^2 at the widest bound the group law uses
  }
}
}  // namespace

// sum_i (+-) P_i with the hot-path group law: points in the key's wire form (affine, x 2^256; 64 B G1 / 128 B G2), signs[i] != 0
// subtracts.  The list is cut in two halves that are accumulated apart (add_mixed29) and then added (add_full29); with
// `twice` the total is added to itself (the doubling branch of add_full29).  out: standard-form affine; returns 1 for infinity.
template <class F> static int chain29(const uint8_t *pts, const uint8_t *signs, size_t n, int twice, uint8_t *out) {
  using C = typename CoordOf<F>::C;
  const size_t pb = sizeof(Affine<F>);
  XYZZ29<C> half[2] = {XYZZ29<C>::inf(), XYZZ29<C>::inf()};
  for (size_t i = 0; i < n; i++) {
    Affine<F> p;
    memcpy(&p, pts + i * pb, pb);
    if (p.is_inf()) continue;
    Affine<F> q{radix_to_261(p.x), radix_to_261(p.y)};
    XYZZ29<C> &acc = half[i >= n / 2];
    acc = add_mixed29<C>(acc, unpack_affine(q), signs[i] != 0);
    acc = unpack_xyzz(pack_xyzz<F>(acc));   // through the packed form, as the kernels store buckets
  }
  XYZZ29<C> tot = add_full29<C>(half[0], half[1]);
  if (twice) tot = add_full29<C>(tot, unpack_xyzz(pack_xyzz<F>(tot)));
  XYZZ<F> res = xyzz_to_256(pack_xyzz<F>(tot));
  if (res.is_inf()) return 1;
  Affine<F> a = to_affine(res);
  if constexpr (sizeof(F) == 32) store_g1_std(out, *reinterpret_cast<G1Affine *>(&a));
  else store_g2_std(out, *reinterpret_cast<G2Affine *>(&a));
  return 0;
}
// 2^(c l) P for l = 1..levels as msm_precompute_kernel computes them: Jacobian doubling chain on the 29-bit limbs, ONE
// inversion of the product of the denominators, affine results.  p: wire form (x 2^256); out: levels x standard-form affine
template <class F> static void jac_levels29(const uint8_t *pt, int c, int levels, uint8_t *out) {
  using C = typename CoordOf<F>::C;
  Affine<F> p;
  memcpy(&p, pt, sizeof p);
  p.x = radix_to_261(p.x); p.y = radix_to_261(p.y);
  Jac29<C> q{C::template unpack<2>(p.x).template to<JX>(), C::template unpack<2>(p.y).template to<JY>(), C::one().template to<JZ>()};
  std::vector<Jac29<C>> lv;
  auto prod = C::one().template to<4>();
  for (int l = 0; l < levels; l++) {
    for (int b = 0; b < c; b++) q = dbl_jac29<C>(q);
    lv.push_back(q);
    prod = mul(prod, q.z).template to<4>();
  }
  auto inv = inv29(prod);
  for (int l = levels - 1; l >= 0; l--) {
    auto pre = C::one().template to<4>();
    for (int j = 0; j < l; j++) pre = mul(pre, lv[j].z).template to<4>();
    auto iz = mul(inv, pre);
    inv = mul(inv, lv[l].z).template to<4>();
    auto iz2 = sqr(iz);
    auto iz3 = mul(iz2, iz);
    Affine<F> a{radix_to_256(C::template pack<2>(canonical_small(mul(lv[l].x, iz2)))), radix_to_256(C::template pack<2>(canonical_small(mul(lv[l].y, iz3))))};
    if constexpr (sizeof(F) == 32) store_g1_std(out + (size_t)l * 64, *reinterpret_cast<G1Affine *>(&a));
    else store_g2_std(out + (size_t)l * 128, *reinterpret_cast<G2Affine *>(&a));
  }
}

#include <atomic>
#include <thread>
#include "shard_group.hpp"

// ShardGroup on the host: `parts` threads meet at `rounds` barriers; thread `fail_part` (if < parts) aborts the group instead of
// arriving at barrier `fail_round`.  Returns 0 when every thread saw what it must: all barriers passed in step (nobody a
// round ahead), or -- with a failure -- every thread returned false at or after the failing round and none hung.
static int shard_group_selftest(unsigned parts, unsigned rounds, unsigned fail_part, unsigned fail_round) {
  zkr::ShardGroup g;
  g.parts = parts;
  g.vecs.resize(parts);
  std::vector<std::atomic<unsigned>> at(parts);
  for (auto &a : at) a.store(0);
  std::atomic<int> bad{0};
  auto work = [&](unsigned i) {
    for (unsigned r = 0; r < rounds; r++) {
      if (i == fail_part && r == fail_round) { g.abort(); return; }
      at[i].store(r + 1);
      const bool ok = g.barrier();
      if (!ok) { if (fail_part >= parts || r < fail_round) bad++; return; }  // a refusal without (or before) the failure
      for (unsigned j = 0; j < parts; j++)
        if (at[j].load() < r + 1) bad++;                                     // somebody had not arrived: the barrier let us through early
    }
    if (fail_part < parts && fail_round < rounds) bad++;                     // ran to the end although a shard failed
  };
  std::vector<std::thread> thr;
  for (unsigned i = 1; i < parts; i++) thr.emplace_back(work, i);
  work(0);
  for (auto &t : thr) t.join();
  return bad.load();
}

extern "C" {
int zkr_host_shard_group_selftest(unsigned parts, unsigned rounds, unsigned fail_part, unsigned fail_round) {
  return shard_group_selftest(parts, rounds, fail_part, fail_round);
}
// field 0 = Fq, 1 = Fr; ops: see f29_op above; inputs / outputs standard form
void zkt29_fp(int field, int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
  if (field == 0) f29_op<Fq29>(op, a, b, out); else f29_op<Fr29>(op, a, b, out);
}
// pack / unpack round trip and the change of radix both ways on a raw 256-bit word string (x 2^256 form in, same out)
void zkt29_radix_roundtrip(const uint8_t *in32, uint8_t *out32, uint8_t *mid32) {
  Fq w;
  memcpy(w.v, in32, 32);
  Fq m = radix_to_261(w);
  memcpy(mid32, m.v, 32);
  Fq back = radix_to_256(m);
  memcpy(out32, back.v, 32);
}
// the four product forms on raw limbs (field29.hpp f29_raw_form, host build = the defining recursion); layout as
// zkr_selftest_f29_forms
void zkt29_raw_forms(int field, int form, const uint32_t *records, size_t n, uint32_t *out) {
  for (size_t i = 0; i < n; i++) {
    uint32_t op[8][9], r[9];
    memcpy(op, records + i * 72, sizeof op);
    if (field == 0) f29_raw_form<Fq29>(form, op, r); else f29_raw_form<Fr29>(form, op, r);
    memcpy(out + i * 9, r, sizeof r);
  }
}
// barrett() and canonical_small() of field29.hpp on raw limbs: in = 9 limbs (low eight below 2^29, value below 64 p),
// out = 9 limbs of barrett(in), canon = 9 limbs of canonical_small(barrett(in))
void zkt29_barrett(int field, const uint32_t *in, uint32_t *out, uint32_t *canon) {
  if (field == 0) {
    L29<Fq29, 128> a; memcpy(a.v, in, 36);
    auto b = barrett(a); memcpy(out, b.v, 36);
    auto c = canonical_small(b); memcpy(canon, c.v, 36);
  } else {
    L29<Fr29, 128> a; memcpy(a.v, in, 36);
    auto b = barrett(a); memcpy(out, b.v, 36);
    auto c = canonical_small(b); memcpy(canon, c.v, 36);
  }
}
// the sweep-less differences on raw limbs (a, b: 9 normalised limbs, value below 13 / 4 half moduli as the accumulator's X / Y
// are): nl = neg_loose(b), sl = sub_loose(a, b) as limbs, for the range and value checks of tests/test_host_arith.py
void zkt29_loose(int field, const uint32_t *a9, const uint32_t *b9, uint32_t *nl, uint32_t *sl, int *k_neg, int *k_sub) {
  if (field == 0) {
    L29<Fq29, 4> a; L29<Fq29, 13> b; memcpy(a.v, a9, 36); memcpy(b.v, b9, 36);
    auto n = neg_loose(b); auto d = sub_loose(a, b);
    memcpy(nl, n.v, 36); memcpy(sl, d.v, 36); *k_neg = decltype(n)::bound / 2; *k_sub = (decltype(d)::bound - 4) / 2;
  } else {
    L29<Fr29, 4> a; L29<Fr29, 13> b; memcpy(a.v, a9, 36); memcpy(b.v, b9, 36);
    auto n = neg_loose(b); auto d = sub_loose(a, b);
    memcpy(nl, n.v, 36); memcpy(sl, d.v, 36); *k_neg = decltype(n)::bound / 2; *k_sub = (decltype(d)::bound - 4) / 2;
  }
}
void zkt29_g1_levels(const uint8_t *pt, int c, int levels, uint8_t *out) { jac_levels29<Fq>(pt, c, levels, out); }
void zkt29_g2_levels(const uint8_t *pt, int c, int levels, uint8_t *out) { jac_levels29<Fq2>(pt, c, levels, out); }
int zkt29_g1_chain(const uint8_t *pts, const uint8_t *signs, size_t n, int twice, uint8_t *out) { return chain29<Fq>(pts, signs, n, twice, out); }
int zkt29_g2_chain(const uint8_t *pts, const uint8_t *signs, size_t n, int twice, uint8_t *out) { return chain29<Fq2>(pts, signs, n, twice, out); }

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
// The verifier's inversion-free Miller loop (projective steps for arbitrary second arguments, prepared lines for fixed ones)
// against the affine loop, after the final exponentiation: `count` rounds of three pairs (a_i G1, b_i G2) with pseudo-random
// small multiples of the generators (g1 64 B, g2 128 B standard form), every split of the pairs into "arbitrary" and "prepared",
// one pair with a point at infinity in the last round.  Returns the number of agreeing comparisons (count * 4 expected).
int zkt_miller_loops_agree(const uint8_t *g1, const uint8_t *g2, uint64_t seed, int count) {
  using namespace zkr::pairing;
  uint64_t st = seed ? seed : 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
  G1Affine P{to_mont(load_fp<FqParams>(g1)), to_mont(load_fp<FqParams>(g1 + 32))};
  G2Affine Q{Fq2{to_mont(load_fp<FqParams>(g2)), to_mont(load_fp<FqParams>(g2 + 32))}, Fq2{to_mont(load_fp<FqParams>(g2 + 64)), to_mont(load_fp<FqParams>(g2 + 96))}};
  auto g2_mul = [&](uint32_t k) {
    G2XYZZ acc = G2XYZZ::inf();
    for (int b = 31; b >= 0; b--) { acc = dbl_xyzz(acc); if ((k >> b) & 1) acc = add_mixed(acc, Q); }
    return to_affine(acc);
  };
  int agree = 0;
  for (int it = 0; it < count; it++) {
    G1Affine ps[3];
    G2Affine qs[3];
    for (int i = 0; i < 3; i++) {
      ps[i] = to_affine(mul_small(to_xyzz(P), (rnd() & 0xffff) + 1));
      qs[i] = g2_mul((rnd() & 0xfffff) + 1);
    }
    if (it == count - 1) ps[1] = G1Affine{Fq::zero(), Fq::one()};  // infinity as the wire form has it: the pair contributes 1
    bool ok = true;
    const Fq12 want = final_exponentiation(multi_miller_loop(ps, qs, 3, &ok));
    if (!ok) continue;
    for (int nfixed = 0; nfixed <= 3; nfixed++) {  // the last nfixed pairs prepared
      G2Prepared prep[3];
      const G2Prepared *pp[3];
      bool good = true;
      for (int j = 0; j < nfixed; j++) { good = g2_prepare(qs[3 - nfixed + j], prep[j]) && good; pp[j] = &prep[j]; }
      bool ok2 = true;
      const Fq12 got = final_exponentiation(miller_loop_mixed(ps, qs, 3 - nfixed, ps + 3 - nfixed, pp, nfixed, &ok2));
      agree += good && ok2 && got == want;
    }
  }
  return agree;
}
// the two G2 membership tests on one twist point (128 B standard form: x.re, x.im, y.re, y.im): bit 0 = the definition
// ([r]Q == infinity), bit 1 = the endomorphism test the verifier runs (psi(Q) == [6x^2]Q), bit 2 = the point is on the twist
int zkt_g2_membership(const uint8_t *pt) {
  using namespace zkr::pairing;
  G2Affine q{Fq2{to_mont(load_fp<FqParams>(pt)), to_mont(load_fp<FqParams>(pt + 32))}, Fq2{to_mont(load_fp<FqParams>(pt + 64)), to_mont(load_fp<FqParams>(pt + 96))}};
  return (g2_in_subgroup_plain(q) ? 1 : 0) | (g2_in_subgroup(q) ? 2 : 0) | (g2_on_curve(q) ? 4 : 0);
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
