// hostops.hpp -- small host-side group operations shared by the proof assembly (zkr_api.hip) and the
// CPU arithmetic test shim.  Uses the same field/curve templates as the device code.
#pragma once
#include <string.h>
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

}  // namespace zkr
