// curve29.hpp -- the XYZZ group law of curve.hpp over the lazily reduced 29-bit-limb fields of field29.hpp: what the bucket
// accumulation and reduction kernels of the five MSMs run (call site
// /root/reference/operator/src/snarks/common.ts:29).  Same formulas, same special cases; every intermediate carries its
// value bound in its type, and the bounds below are checked by the compiler (a formula that could overflow R / p = 169
// does not build).  In memory points keep the layouts of curve.hpp (8 x 32-bit words per coordinate): Affine<F> /
// XYZZ<F> with F = Fq, Fq2 are the PACKED forms, coordinates there are x * 2^261 mod p (tables: canonical; buckets:
// below 2 p).
#pragma once
#include "curve.hpp"
#include "field29.hpp"

namespace zkr {

// coordinate families: G1 over Fq, G2 over Fq2
struct G1C {
  template <int H> using T = L29<Fq29, H>;
  using W = Fq;  // packed form of one coordinate
  template <int H> static ZKR_HD T<H> unpack(const Fq &w) { return unpack29<Fq29, H>(w.v); }
  template <int H> static ZKR_HD Fq pack(const T<H> &a) { Fq r; pack29(a, r.v); return r; }
  static ZKR_HD T<2> one() { return one29<Fq29>(); }
  static ZKR_HD T<2> to256() { return const29<Fq29>(Fq29::TO256); }
  static ZKR_HD T<2> to261() { return const29<Fq29>(Fq29::TO261); }
};
struct G2C {
  template <int H> using T = Q29<H>;
  using W = Fq2;
  template <int H> static ZKR_HD T<H> unpack(const Fq2 &w) { return Q29<H>{unpack29<Fq29, H>(w.a.v), unpack29<Fq29, H>(w.b.v)}; }
  template <int H> static ZKR_HD Fq2 pack(const T<H> &x) { Fq2 r; pack29(x.a, r.a.v); pack29(x.b, r.b.v); return r; }
  static ZKR_HD T<2> one() { return Q29<2>{one29<Fq29>(), L29<Fq29, 2>::zero()}; }
  static ZKR_HD T<2> to256() { return Q29<2>{const29<Fq29>(Fq29::TO256), L29<Fq29, 2>::zero()}; }
  static ZKR_HD T<2> to261() { return Q29<2>{const29<Fq29>(Fq29::TO261), L29<Fq29, 2>::zero()}; }
};
template <class F> struct CoordOf;
template <> struct CoordOf<Fq> { using C = G1C; };
template <> struct CoordOf<Fq2> { using C = G2C; };

// bounds (in half moduli) of the coordinates of a point held in registers: X comes out of two subtractions
// (R^2 - P^3 - 2 X1 P^2), the others out of products
constexpr int HX = 13, HY = 4;

template <class C>
struct Affine29 {
  typename C::template T<2> x, y;  // canonical table values
};
template <class C>
struct XYZZ29 {
  typename C::template T<HX> x;
  typename C::template T<HY> y, zz, zzz;
  static ZKR_HD XYZZ29 inf() {
    XYZZ29 r;
    r.x = decltype(r.x)::zero(); r.y = decltype(r.y)::zero(); r.zz = decltype(r.zz)::zero(); r.zzz = decltype(r.zzz)::zero();
    return r;
  }
  ZKR_HD bool is_inf() const { return zz.all_zero(); }  // infinity is always written as exact zeros; a finite point's ZZ is never 0 mod p
};
template <class C, class X, class Y, class ZZ, class ZZZ>
ZKR_HD XYZZ29<C> make_xyzz(const X &x, const Y &y, const ZZ &zz, const ZZZ &zzz) {
  XYZZ29<C> r;
  r.x = x.template to<HX>(); r.y = y.template to<HY>(); r.zz = zz.template to<HY>(); r.zzz = zzz.template to<HY>();
  return r;
}

// ---- packed <-> registers
template <class F>
ZKR_HD Affine29<typename CoordOf<F>::C> unpack_affine(const Affine<F> &p) {
  using C = typename CoordOf<F>::C;
  return Affine29<C>{C::template unpack<2>(p.x), C::template unpack<2>(p.y)};
}
template <class F>
ZKR_HD XYZZ29<typename CoordOf<F>::C> unpack_xyzz(const XYZZ<F> &p) {  // stored coordinates are below 2 p (pack_xyzz)
  using C = typename CoordOf<F>::C;
  XYZZ29<C> r;
  r.x = C::template unpack<HY>(p.x).template to<HX>();
  r.y = C::template unpack<HY>(p.y); r.zz = C::template unpack<HY>(p.zz); r.zzz = C::template unpack<HY>(p.zzz);
  return r;
}
template <class F>
ZKR_HD XYZZ<F> pack_xyzz(const XYZZ29<typename CoordOf<F>::C> &p) {
  using C = typename CoordOf<F>::C;
  if (p.is_inf()) return XYZZ<F>::inf();
  return XYZZ<F>{C::template pack<3>(weak(p.x)), C::template pack<HY>(p.y), C::template pack<HY>(p.zz), C::template pack<HY>(p.zzz)};
}

// 2 * (affine q), y possibly negated already
template <class C, class QX, class QY>
ZKR_HD XYZZ29<C> dbl_affine29(const QX &qx, const QY &qy) {
  auto u = dbl(qy);
  auto v = sqr(u);
  auto w = mul(u, v);
  auto s = mul(qx, v);
  auto xx = sqr(qx);
  auto m = add(dbl(xx), xx);
  auto x3 = sub(sqr(m), dbl(s));
  auto y3 = mul_sub(m, sub_factor(s, x3), w, qy);
  return make_xyzz<C>(x3, y3, v, w);
}

template <class C>
ZKR_HD XYZZ29<C> dbl_xyzz29(const XYZZ29<C> &p) {
  if (p.is_inf()) return p;
  auto u = dbl(p.y);
  auto v = sqr(u);
  auto w = mul(u, v);
  auto s = mul(p.x, v);
  auto xx = sqr(p.x);
  auto m = add(dbl(xx), xx);
  auto x3 = sub(sqr(m), dbl(s));
  auto y3 = mul_sub(m, sub_factor(s, x3), w, p.y);
  return make_xyzz<C>(x3, y3, mul(v, p.zz), mul(w, p.zzz));
}

// The products of the group law in the order they are WRITTEN (a scheduling barrier after each): every product consumes operands the
// next ones no longer need, so at most eight coordinates are alive at any point.  Left to itself the compiler interleaves the
// independent products (each is one opaque asm statement: nothing to gain) and needs 30-150 VGPRs more -- registers that decide
// which kernels can share a SIMD with the accumulations (kernels_msm.hpp, "built to FIT").
#if defined(__HIP_DEVICE_COMPILE__)
#define ZKR_PIN_ORDER() __builtin_amdgcn_sched_barrier(0)  // the products below in THIS order: each consumes operands the next no longer needs
#else
#define ZKR_PIN_ORDER() ((void)0)
#endif
// acc + q (q affine, not infinity: callers filter); neg_q adds -q.  The sign goes into the product Y2 ZZZ1 as a factor
// without its carry sweep (field29.hpp U29), as do Q - X3 and -Y1 in the Y coordinate: three sweeps and a select less per
// addition; the negated y itself is only formed on the rare paths that store it.
template <class C>
ZKR_HD XYZZ29<C> add_mixed29(const XYZZ29<C> &acc, const Affine29<C> &q, bool neg_q) {
  if (acc.is_inf()) {
    auto qy = q.y;
    if (neg_q) qy = neg(q.y).template to<2>();  // p - y (<= p)
    return make_xyzz<C>(q.x, qy, C::one(), C::one());
  }
  auto u2 = mul(q.x, acc.zz);
  ZKR_PIN_ORDER();
  auto s2 = mul_cneg(q.y, neg_q, acc.zzz);
  ZKR_PIN_ORDER();
  auto p = sub(u2, acc.x);
  auto r = sub(s2, acc.y);
  auto pp = sqr(p);
  if (maybe_zero_mod_p(pp) && is_zero_mod_p(pp)) {  // same x: the same point (double it) or its negative (infinity)
    if (is_zero_mod_p(sqr(r))) {
      auto qy = q.y;
      if (neg_q) qy = neg(q.y).template to<2>();
      return dbl_affine29<C>(q.x, qy);
    }
    return XYZZ29<C>::inf();
  }
  ZKR_PIN_ORDER();
  auto ppp = mul(p, pp);
  ZKR_PIN_ORDER();
  auto qq = mul(acc.x, pp);
  ZKR_PIN_ORDER();
  auto zz3 = mul(acc.zz, pp);
  ZKR_PIN_ORDER();
  auto zzz3 = mul(acc.zzz, ppp);
  ZKR_PIN_ORDER();
  auto x3 = sub_sub_dbl(sqr(r), ppp, qq);  // R^2 - P^3 - 2 Q, one carry sweep
  auto y3 = mul_sub(r, sub_factor(qq, x3), acc.y, ppp);
  return make_xyzz<C>(x3, y3, zz3, zzz3);
}

// a + b, both XYZZ.  `again()` yields b once more (from the memory it came from): the one case that needs a whole operand after
// the first products -- a == b, the doubling -- fetches it again instead of keeping its X and Y in registers through every
// addition, and the products of the ZZ / ZZZ pairs are formed first so that the four inputs die early.  In the kernels whose
// register count decides where they can run (the bucket reductions beside the accumulations, kernels_msm.hpp) that is 30-60 VGPRs.
template <class C, class Again>
ZKR_HD XYZZ29<C> add_full29(const XYZZ29<C> &a, const XYZZ29<C> &b, Again again) {
  if (a.is_inf()) return b;
  if (b.is_inf()) return a;
  // at most eight coordinates alive at any point (the compiler's own order interleaves the independent products and needs
  // 30-60 registers more)
  auto u1 = mul(a.x, b.zz);
  ZKR_PIN_ORDER();
  auto u2 = mul(b.x, a.zz);
  ZKR_PIN_ORDER();
  auto s1 = mul(a.y, b.zzz);
  ZKR_PIN_ORDER();
  auto s2 = mul(b.y, a.zzz);
  ZKR_PIN_ORDER();
  auto zzp = mul(a.zz, b.zz);
  ZKR_PIN_ORDER();
  auto zzzp = mul(a.zzz, b.zzz);
  ZKR_PIN_ORDER();
  auto p = sub(u2, u1);
  auto r = sub(s2, s1);
  auto pp = sqr(p);
  if (maybe_zero_mod_p(pp) && is_zero_mod_p(pp)) {
    if (is_zero_mod_p(sqr(r))) return dbl_xyzz29(again());
    return XYZZ29<C>::inf();
  }
  ZKR_PIN_ORDER();
  auto ppp = mul(p, pp);
  ZKR_PIN_ORDER();
  auto qq = mul(u1, pp);
  ZKR_PIN_ORDER();
  auto zz3 = mul(zzp, pp);
  ZKR_PIN_ORDER();
  auto zzz3 = mul(zzzp, ppp);
  ZKR_PIN_ORDER();
  auto x3 = sub_sub_dbl(sqr(r), ppp, qq);
  auto y3 = mul_sub(r, sub_factor(qq, x3), s1, ppp);
  return make_xyzz<C>(x3, y3, zz3, zzz3);
}
template <class C>
ZKR_HD XYZZ29<C> add_full29(const XYZZ29<C> &a, const XYZZ29<C> &b) {
  return add_full29<C>(a, b, [&]() { return b; });
}

// ---- Jacobian doubling chain of the window-table build (key load; kernels_msm.hpp msm_precompute_kernel).  (X, Y, Z) with
// x = X / Z^2, y = Y / Z^3: ONE denominator per point, so the 12 multiples a table stores per base point are made affine
// with one inversion (Montgomery's trick over the levels) instead of twelve.  a = 0 (dbl-2009-l, with 4 X Y^2 as a product):
// 3 products, 4 squares, two quotient-estimate reductions; bounds in and out: X, Y <= 5, Z <= 8 half moduli.
constexpr int JX = 5, JY = 5, JZ = 8;
template <class C>
struct Jac29 {
  typename C::template T<JX> x;
  typename C::template T<JY> y;
  typename C::template T<JZ> z;
};
template <class C>
ZKR_HD Jac29<C> dbl_jac29(const Jac29<C> &p) {
  auto a = sqr(p.x);                                          // X^2
  auto b = sqr(p.y);                                          // Y^2
  auto c8 = dbl(dbl(dbl(sqr(b))));                            // 8 Y^4
  auto s = dbl(dbl(mul(p.x, b)));                             // 4 X Y^2
  auto e = add(dbl(a), a);                                    // 3 X^2
  auto x3 = barrett(sub(sqr(e), dbl(s)));                     // E^2 - 2 S
  auto y3 = barrett(sub(mul(e, sub(s, x3)), c8));             // E (S - X3) - 8 Y^4
  auto z3 = dbl(mul(p.y, p.z));                               // 2 Y Z
  return Jac29<C>{x3.template to<JX>(), y3.template to<JY>(), z3.template to<JZ>()};
}

// ---- change of Montgomery radix at the edges of the hot path (cold)
// packed coordinate x 2^256 (key material, field.hpp) -> packed canonical x 2^261, and back
template <class F>
ZKR_HD_COLD F radix_to_261(const F &w) {
  using C = typename CoordOf<F>::C;
  return C::template pack<2>(canonical(mul(C::template unpack<10>(w), C::to261())));
}
template <class F>
ZKR_HD_COLD F radix_to_256(const F &w) {
  using C = typename CoordOf<F>::C;
  return C::template pack<2>(canonical(mul(C::template unpack<10>(w), C::to256())));
}
// The same ON coordinates in memory (global or LDS): the non-inlined call takes a pointer in registers, where the by-value forms
// above put their 32 / 64-byte argument and result on the stack -- the only scratch the kernels that call them once per thread
// had (msm_reduce3_kernel 208 / 464 B per lane, msm_precompute_kernel 128 / 320 B, radix_convert_kernel 128 / 320 B).
// (canonical_small, the inline form, instead of the out-of-line canonical(): a product is below 2 p, one conditional subtraction;
// a nested out-of-line call would bring the stack frame back.)
template <class F>
ZKR_HD_COLD void radix_to_261_at(F *w) {
  using C = typename CoordOf<F>::C;
  *w = C::template pack<2>(canonical_small(mul(C::template unpack<10>(*w), C::to261())));
}
template <class F>
ZKR_HD_COLD void radix_to_256_at(F *w) {
  using C = typename CoordOf<F>::C;
  *w = C::template pack<2>(canonical_small(mul(C::template unpack<10>(*w), C::to256())));
}
template <class F>
ZKR_HD void xyzz_to_256_at(XYZZ<F> *p) {  // inline: the four leaf calls are made from the kernel itself (a nested call would save its return address on the stack)
  if (p->is_inf()) return;
  radix_to_256_at(&p->x); radix_to_256_at(&p->y); radix_to_256_at(&p->zz); radix_to_256_at(&p->zzz);
}
template <class F>
ZKR_HD_COLD XYZZ<F> xyzz_to_256(const XYZZ<F> &p) {  // an MSM result on its way to the host assembly (hostops.hpp, field.hpp arithmetic)
  if (p.is_inf()) return p;
  return XYZZ<F>{radix_to_256(p.x), radix_to_256(p.y), radix_to_256(p.zz), radix_to_256(p.zzz)};
}

}  // namespace zkr
