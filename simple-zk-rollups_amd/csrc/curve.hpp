// curve.hpp -- short-Weierstrass (a = 0) group law in XYZZ coordinates, generic over the coordinate
// field: F = Fq gives G1 (y^2 = x^3 + 3), F = Fq2 gives G2 on the twist.  The formulas never use the
// curve constant b, so one template serves both groups.
//
// XYZZ: (X, Y, ZZ, ZZZ) with x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ == 0.
// Mixed add (bucket += affine point) costs 8M + 2S, the cheapest complete-enough form for Pippenger
// bucket accumulation; the rare P == +-Q cases are handled explicitly.
// Affine wire format = the key sections written by binarify.ts:92-102: Montgomery coordinates,
// x == 0 encodes the point at infinity (snarkjs' [0,1,0]; x = 0 is never on either curve).
#pragma once
#include "field.hpp"

namespace zkr {

template <class F>
struct Affine {
  F x, y;
  ZKR_HD bool is_inf() const { return x.is_zero(); }
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  static ZKR_HD XYZZ inf() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
  ZKR_HD bool is_inf() const { return zz.is_zero(); }
};

template <class F>
ZKR_HD XYZZ<F> to_xyzz(const Affine<F> &p) {
  if (p.is_inf()) return XYZZ<F>::inf();
  return XYZZ<F>{p.x, p.y, F::one(), F::one()};
}

// 2*(affine p) -> XYZZ   (dbl-2008-s-1 specialised to Z = 1)
template <class F>
ZKR_HD XYZZ<F> dbl_affine(const Affine<F> &p) {
  F u = dbl(p.y);
  F v = sqr(u);
  F w = mul(u, v);
  F s = mul(p.x, v);
  F xx = sqr(p.x);
  F m = add(dbl(xx), xx);
  F x3 = sub(sqr(m), dbl(s));
  F y3 = mul_sub(m, sub(s, x3), w, p.y);
  return XYZZ<F>{x3, y3, v, w};
}

template <class F>
ZKR_HD XYZZ<F> dbl_xyzz_inl(const XYZZ<F> &p) {
  if (p.is_inf()) return p;
  F u = dbl(p.y);
  F v = sqr(u);
  F w = mul(u, v);
  F s = mul(p.x, v);
  F xx = sqr(p.x);
  F m = add(dbl(xx), xx);
  F x3 = sub(sqr(m), dbl(s));
  F y3 = mul_sub(m, sub(s, x3), w, p.y);
  return XYZZ<F>{x3, y3, mul(v, p.zz), mul(w, p.zzz)};
}

template <class F>
ZKR_HD_COLD XYZZ<F> dbl_xyzz(const XYZZ<F> &p) {
  return dbl_xyzz_inl(p);
}

// acc + q, q affine and NOT infinity (callers filter infinity).  neg_q adds -q instead.
template <class F>
ZKR_HD XYZZ<F> add_mixed(const XYZZ<F> &acc, const Affine<F> &q_in, bool neg_q = false) {
  Affine<F> q = q_in;
  if (neg_q) q.y = neg(q.y);
  if (acc.is_inf()) return XYZZ<F>{q.x, q.y, F::one(), F::one()};
  F u2 = mul(q.x, acc.zz);
  F s2 = mul(q.y, acc.zzz);
  F p = sub(u2, acc.x);
  F r = sub(s2, acc.y);
  if (p.is_zero()) {
    if (r.is_zero()) return dbl_affine(q);
    return XYZZ<F>::inf();
  }
  F pp = sqr(p);
  F ppp = mul(p, pp);
  F qq = mul(acc.x, pp);
  F x3 = sub(sub(sqr(r), ppp), dbl(qq));
  F y3 = mul_sub(r, sub(qq, x3), acc.y, ppp);
  return XYZZ<F>{x3, y3, mul(acc.zz, pp), mul(acc.zzz, ppp)};
}

// full add, both XYZZ (bucket reduction running sums): 12M + 2S; inlined form for the reduction kernels
template <class F>
ZKR_HD XYZZ<F> add_full_inl(const XYZZ<F> &a, const XYZZ<F> &b) {
  if (a.is_inf()) return b;
  if (b.is_inf()) return a;
  F u1 = mul(a.x, b.zz);
  F u2 = mul(b.x, a.zz);
  F s1 = mul(a.y, b.zzz);
  F s2 = mul(b.y, a.zzz);
  F p = sub(u2, u1);
  F r = sub(s2, s1);
  if (p.is_zero()) {
    if (r.is_zero()) return dbl_xyzz_inl(a);
    return XYZZ<F>::inf();
  }
  F pp = sqr(p);
  F ppp = mul(p, pp);
  F qq = mul(u1, pp);
  F x3 = sub(sub(sqr(r), ppp), dbl(qq));
  F y3 = mul_sub(r, sub(qq, x3), s1, ppp);
  return XYZZ<F>{x3, y3, mul(mul(a.zz, b.zz), pp), mul(mul(a.zzz, b.zzz), ppp)};
}

// out-of-line form for cold paths (host assembly, setup): keeps those builds small
template <class F>
ZKR_HD_COLD XYZZ<F> add_full(const XYZZ<F> &a, const XYZZ<F> &b) {
  return add_full_inl(a, b);
}

// k * p for a small unsigned k (bucket-group offsets): left-to-right double-and-add
template <class F>
ZKR_HD_COLD XYZZ<F> mul_small(const XYZZ<F> &p, uint32_t k) {
  XYZZ<F> acc = XYZZ<F>::inf();
  if (k == 0) return acc;
  for (int i = 31 - __builtin_clz(k); i >= 0; i--) {
    acc = dbl_xyzz(acc);
    if ((k >> i) & 1) acc = add_full(acc, p);
  }
  return acc;
}

// XYZZ -> affine (x = X/ZZ, y = Y/ZZZ); infinity maps to x = 0 (the wire encoding), y = one
template <class F>
ZKR_HD_COLD Affine<F> to_affine(const XYZZ<F> &p) {
  if (p.is_inf()) return Affine<F>{F::zero(), F::one()};
  F zi = inv(p.zzz);                       // 1/ZZZ
  F inv_zz = mul(sqr(p.zz), sqr(zi));      // ZZ^3 = ZZZ^2  =>  1/ZZ = ZZ^2/ZZZ^2
  return Affine<F>{mul(p.x, inv_zz), mul(p.y, zi)};
}

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;

}  // namespace zkr
