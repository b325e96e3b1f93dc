// hostops.hpp -- small host-side group operations shared by the proof assembly (zkr_api.hip) and the
// CPU arithmetic test shim.  Uses the same field/curve templates as the device code.
#pragma once
#include <string.h>
#include <vector>
#include "curve.hpp"

namespace zkr {

struct U256 {
  uint32_t v[8];
};
static inline U256 load_u256(const uint8_t *p) { U256 r; memcpy(r.v, p, 32); return r; }

template <class PM>
static inline Fp<PM> load_fp(const uint8_t *p) { Fp<PM> r; memcpy(r.v, p, 32); return r; }
template <class PM>
static inline void store_fp(uint8_t *p, const Fp<PM> &a) { memcpy(p, a.v, 32); }

static inline G1Affine load_g1(const uint8_t *p) { return G1Affine{load_fp<FqParams>(p), load_fp<FqParams>(p + 32)}; }
static inline G2Affine load_g2(const uint8_t *p) {
  return G2Affine{Fq2{load_fp<FqParams>(p), load_fp<FqParams>(p + 32)}, Fq2{load_fp<FqParams>(p + 64), load_fp<FqParams>(p + 96)}};
}

// k * P, k a 256-bit standard-form integer (left-to-right double-and-add)
template <class F>
static inline XYZZ<F> scalar_mul(const XYZZ<F> &p, const U256 &k) {
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int i = 255; i >= 0; i--) {
    acc = dbl_xyzz(acc);
    if ((k.v[i >> 5] >> (i & 31)) & 1) acc = add_full(acc, p);
  }
  return acc;
}

// affine standard-form bytes (x,y[,..]) of a non-infinity point
static inline void store_g1_std(uint8_t *out, const G1Affine &a) {
  store_fp(out, from_mont(a.x));
  store_fp(out + 32, from_mont(a.y));
}
static inline void store_g2_std(uint8_t *out, const G2Affine &a) {
  store_fp(out, from_mont(a.x.a));
  store_fp(out + 32, from_mont(a.x.b));
  store_fp(out + 64, from_mont(a.y.a));
  store_fp(out + 96, from_mont(a.y.b));
}

// ---- proof assembly helpers (host): 4-bit windows over 256-bit standard-form scalars ----------------------------
static inline uint32_t nibble(const U256 &k, int i) { return (k.v[i >> 3] >> ((i & 7) * 4)) & 15; }

// Fixed base (delta_1, delta_2 of a key): tab[15 * i + d - 1] = d * 16^i * P, affine, for i < 64, d = 1..15, so
// k * P is at most 64 mixed additions.  Built once per key.
template <class F>
static inline std::vector<Affine<F>> fixed_base_table(const Affine<F> &p) {
  std::vector<XYZZ<F>> pts;
  pts.reserve(64 * 15);
  XYZZ<F> base = to_xyzz(p);
  for (int i = 0; i < 64; i++) {
    XYZZ<F> acc = base;
    for (int d = 1; d <= 15; d++) {
      pts.push_back(acc);
      acc = add_full(acc, base);
    }
    base = acc;  // 16 * base
  }
  std::vector<Affine<F>> tab(pts.size());
  for (size_t j = 0; j < pts.size(); j++) tab[j] = to_affine(pts[j]);
  return tab;
}
template <class F>
static inline XYZZ<F> fixed_base_mul(const std::vector<Affine<F>> &tab, const U256 &k) {
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int i = 0; i < 64; i++) {
    uint32_t d = nibble(k, i);
    if (d && !tab[15 * i + d - 1].is_inf()) acc = add_mixed(acc, tab[15 * i + d - 1]);
  }
  return acc;
}
// a * P + b * Q with shared doublings (two variable bases: the MSM results the blinding multiplies)
template <class F>
static inline XYZZ<F> double_scalar_mul(const XYZZ<F> &p, const U256 &a, const XYZZ<F> &q, const U256 &b) {
  XYZZ<F> tp[16], tq[16];
  tp[0] = tq[0] = XYZZ<F>::inf();
  for (int d = 1; d < 16; d++) {
    tp[d] = add_full(tp[d - 1], p);
    tq[d] = add_full(tq[d - 1], q);
  }
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int i = 63; i >= 0; i--) {
    if (!acc.is_inf())
      for (int j = 0; j < 4; j++) acc = dbl_xyzz(acc);
    uint32_t da = nibble(a, i), db = nibble(b, i);
    if (da) acc = add_full(acc, tp[da]);
    if (db) acc = add_full(acc, tq[db]);
  }
  return acc;
}

}  // namespace zkr
