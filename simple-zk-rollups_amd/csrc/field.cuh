// field.cuh -- BN254 Fq / Fr Montgomery arithmetic on 8 x u32 limbs (device and host).
//
// Written for the CDNA4 VALU: every product is a 32x32+64 multiply-add (v_mad_u64_u32); loops are
// fully unrolled so an element lives in 8 VGPRs.  No MFMA: this is integer work.
// Moduli as in /root/reference/operator/src/utils/binarify.ts:79-81 (q) and :86-88 (r); Montgomery
// radix 2^256 as in binarify.ts:78-90.  The same code compiles for the host (clang) so that the
// arithmetic can be unit-tested on CPU against the oracle (tests/test_host_arith.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define ZKR_HD __host__ __device__ __forceinline__
// cold group-law helpers are real calls on the device: keeps hot loops small and builds fast
#define ZKR_HD_COLD inline __host__ __device__ __noinline__
#else
#define ZKR_HD inline __attribute__((always_inline))
#define ZKR_HD_COLD inline
#endif

namespace zkr {

struct FqParams {
  // q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
  static constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                    0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t INV = 0xe4866389u;  // -q^-1 mod 2^32
  static constexpr uint32_t R1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                     0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                     0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};
struct FrParams {
  // r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
  static constexpr uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                    0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t INV = 0xefffffffu;  // -r^-1 mod 2^32
  static constexpr uint32_t R1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                     0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                     0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

template <class PM>
struct Fp {
  uint32_t v[8];

  static ZKR_HD Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
  }
  static ZKR_HD Fp one() {  // Montgomery 1
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = PM::R1[i];
    return r;
  }
  static ZKR_HD Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = PM::R2[i];
    return r;
  }
  ZKR_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= v[i];
    return o == 0;
  }
  ZKR_HD bool operator==(const Fp &b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= v[i] ^ b.v[i];
    return o == 0;
  }
};

// r = a - p if a >= p else a   (a < 2p)
template <class PM>
ZKR_HD Fp<PM> reduce_once(const Fp<PM> &a) {
  Fp<PM> d;
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], PM::P[i], br, &br);
  Fp<PM> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = br ? a.v[i] : d.v[i];
  return r;
}

template <class PM>
ZKR_HD Fp<PM> add(const Fp<PM> &a, const Fp<PM> &b) {
  Fp<PM> s;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
  return reduce_once(s);  // p < 2^254 so a+b < 2^255: no carry out
}

template <class PM>
ZKR_HD Fp<PM> sub(const Fp<PM> &a, const Fp<PM> &b) {
  Fp<PM> d;
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], b.v[i], br, &br);
  uint32_t mask = 0u - br, c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d.v[i] = __builtin_addc(d.v[i], PM::P[i] & mask, c, &c);
  return d;
}

template <class PM>
ZKR_HD Fp<PM> neg(const Fp<PM> &a) {
  if (a.is_zero()) return a;
  Fp<PM> d;
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(PM::P[i], a.v[i], br, &br);
  return d;
}

template <class PM>
ZKR_HD Fp<PM> dbl(const Fp<PM> &a) {
  return add(a, a);
}

// Montgomery product a*b/2^256 mod p, CIOS with the two inner loops fused (valid because the top
// bit of p is clear, so the running sum never needs a ninth limb).  136 multiply-adds.
template <class PM>
ZKR_HD Fp<PM> mul(const Fp<PM> &a, const Fp<PM> &b) {
  uint32_t t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t A = (uint64_t)a.v[0] * b.v[i] + t[0];
    uint32_t m = (uint32_t)A * PM::INV;
    uint64_t C = (uint64_t)m * PM::P[0] + (uint32_t)A;
    A >>= 32;
    C >>= 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      A += (uint64_t)a.v[j] * b.v[i] + t[j];
      C += (uint64_t)m * PM::P[j] + (uint32_t)A;
      t[j - 1] = (uint32_t)C;
      A >>= 32;
      C >>= 32;
    }
    t[7] = (uint32_t)(A + C);
  }
  Fp<PM> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return reduce_once(r);
}

template <class PM>
ZKR_HD Fp<PM> sqr(const Fp<PM> &a) {
  return mul(a, a);
}

template <class PM>
ZKR_HD Fp<PM> to_mont(const Fp<PM> &a) {
  return mul(a, Fp<PM>::r2());
}
template <class PM>
ZKR_HD Fp<PM> from_mont(const Fp<PM> &a) {
  Fp<PM> one;
#pragma unroll
  for (int i = 0; i < 8; i++) one.v[i] = i == 0;
  return mul(a, one);
}

// a^(p-2): used only off the hot path (affine conversion in the setup kernels)
template <class PM>
ZKR_HD_COLD Fp<PM> inv(const Fp<PM> &a) {
  Fp<PM> r = Fp<PM>::one(), b = a;
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = PM::P[i];
  e[0] -= 2;  // low limb of both moduli is > 2
  for (int i = 0; i < 254; i++) {
    if ((e[i >> 5] >> (i & 31)) & 1) r = mul(r, b);
    b = sqr(b);
  }
  return r;
}

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

// ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1)
struct Fq2 {
  Fq a, b;  // a + b*u
  static ZKR_HD Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
  static ZKR_HD Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
  ZKR_HD bool is_zero() const { return a.is_zero() && b.is_zero(); }
  ZKR_HD bool operator==(const Fq2 &o) const { return a == o.a && b == o.b; }
};
ZKR_HD Fq2 add(const Fq2 &x, const Fq2 &y) { return Fq2{add(x.a, y.a), add(x.b, y.b)}; }
ZKR_HD Fq2 sub(const Fq2 &x, const Fq2 &y) { return Fq2{sub(x.a, y.a), sub(x.b, y.b)}; }
ZKR_HD Fq2 neg(const Fq2 &x) { return Fq2{neg(x.a), neg(x.b)}; }
ZKR_HD Fq2 dbl(const Fq2 &x) { return Fq2{dbl(x.a), dbl(x.b)}; }
ZKR_HD Fq2 mul(const Fq2 &x, const Fq2 &y) {  // Karatsuba, 3 Fq products
  Fq t0 = mul(x.a, y.a), t1 = mul(x.b, y.b);
  Fq m = mul(add(x.a, x.b), add(y.a, y.b));
  return Fq2{sub(t0, t1), sub(sub(m, t0), t1)};
}
ZKR_HD Fq2 sqr(const Fq2 &x) {  // (a+b)(a-b) + 2ab u, 2 Fq products
  Fq t = mul(x.a, x.b);
  return Fq2{mul(add(x.a, x.b), sub(x.a, x.b)), dbl(t)};
}
ZKR_HD_COLD Fq2 inv(const Fq2 &x) {
  Fq n = inv(add(sqr(x.a), sqr(x.b)));
  return Fq2{mul(x.a, n), neg(mul(x.b, n))};
}

}  // namespace zkr
