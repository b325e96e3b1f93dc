// field.hpp -- BN254 Fq / Fr Montgomery arithmetic on 8 x u32 limbs (device and host).
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

#if defined(__HIP_DEVICE_COMPILE__)
// 96-bit column accumulator (c2 : lo) += sum_k x_k * y_k (device only; the host uses mul_host64 below).  On gfx950 every term is exactly two VALU
// instructions: v_mad_u64_u32 adds the 64-bit product into the low pair and leaves the carry in VCC,
// v_addc_co_u32 folds it into the third word -- no register shuffling between multiply-adds (the
// compiler's own lowering of the 64-bit carry chain costs ~3 v_mov + one 64-bit add per multiply-add).
// One asm statement per column half keeps hipcc's per-statement s_nop pad off the critical path.
// Pure VALU, no memory operands, VCC declared clobbered (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void mac96_1(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "v"(y0) : "vcc");
}
template <uint32_t Y0>
__device__ __forceinline__ void mac96c_1(uint64_t &lo, uint32_t &c2, uint32_t x0) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "s"(Y0) : "vcc");
}
__device__ __forceinline__ void mac96_2(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "vcc");
}
template <uint32_t Y0, uint32_t Y1>
__device__ __forceinline__ void mac96c_2(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t x1) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "s"(Y0), "v"(x1), "s"(Y1) : "vcc");
}
__device__ __forceinline__ void mac96_3(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2) : "vcc");
}
template <uint32_t Y0, uint32_t Y1, uint32_t Y2>
__device__ __forceinline__ void mac96c_3(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t x1, uint32_t x2) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "s"(Y0), "v"(x1), "s"(Y1), "v"(x2), "s"(Y2) : "vcc");
}
__device__ __forceinline__ void mac96_4(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3) : "vcc");
}
template <uint32_t Y0, uint32_t Y1, uint32_t Y2, uint32_t Y3>
__device__ __forceinline__ void mac96c_4(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "s"(Y0), "v"(x1), "s"(Y1), "v"(x2), "s"(Y2), "v"(x3), "s"(Y3) : "vcc");
}
__device__ __forceinline__ void mac96_5(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4) : "vcc");
}
template <uint32_t Y0, uint32_t Y1, uint32_t Y2, uint32_t Y3, uint32_t Y4>
__device__ __forceinline__ void mac96c_5(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "s"(Y0), "v"(x1), "s"(Y1), "v"(x2), "s"(Y2), "v"(x3), "s"(Y3), "v"(x4), "s"(Y4) : "vcc");
}
__device__ __forceinline__ void mac96_6(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4, uint32_t x5, uint32_t y5) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %12, %13, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5) : "vcc");
}
template <uint32_t Y0, uint32_t Y1, uint32_t Y2, uint32_t Y3, uint32_t Y4, uint32_t Y5>
__device__ __forceinline__ void mac96c_6(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t x5) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %12, %13, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "s"(Y0), "v"(x1), "s"(Y1), "v"(x2), "s"(Y2), "v"(x3), "s"(Y3), "v"(x4), "s"(Y4), "v"(x5), "s"(Y5) : "vcc");
}
__device__ __forceinline__ void mac96_7(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4, uint32_t x5, uint32_t y5, uint32_t x6, uint32_t y6) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %12, %13, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %14, %15, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5), "v"(x6), "v"(y6) : "vcc");
}
template <uint32_t Y0, uint32_t Y1, uint32_t Y2, uint32_t Y3, uint32_t Y4, uint32_t Y5, uint32_t Y6>
__device__ __forceinline__ void mac96c_7(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t x5, uint32_t x6) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %12, %13, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %14, %15, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "s"(Y0), "v"(x1), "s"(Y1), "v"(x2), "s"(Y2), "v"(x3), "s"(Y3), "v"(x4), "s"(Y4), "v"(x5), "s"(Y5), "v"(x6), "s"(Y6) : "vcc");
}
__device__ __forceinline__ void mac96_8(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4, uint32_t x5, uint32_t y5, uint32_t x6, uint32_t y6, uint32_t x7, uint32_t y7) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %12, %13, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %14, %15, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %16, %17, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5), "v"(x6), "v"(y6), "v"(x7), "v"(y7) : "vcc");
}
template <uint32_t Y0, uint32_t Y1, uint32_t Y2, uint32_t Y3, uint32_t Y4, uint32_t Y5, uint32_t Y6, uint32_t Y7>
__device__ __forceinline__ void mac96c_8(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t x5, uint32_t x6, uint32_t x7) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %12, %13, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %14, %15, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %16, %17, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(c2) : "v"(x0), "s"(Y0), "v"(x1), "s"(Y1), "v"(x2), "s"(Y2), "v"(x3), "s"(Y3), "v"(x4), "s"(Y4), "v"(x5), "s"(Y5), "v"(x6), "s"(Y6), "v"(x7), "s"(Y7) : "vcc");
}

// first multiply-adds of a column: the carry word starts from the carry itself (no separate clearing of c2)
__device__ __forceinline__ void mac96f_1(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc" : "+v"(lo), "=&v"(c2) : "v"(x0), "v"(y0) : "vcc");
}
__device__ __forceinline__ void mac96f_2(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "=&v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "vcc");
}
__device__ __forceinline__ void mac96f_3(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "=&v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2) : "vcc");
}
__device__ __forceinline__ void mac96f_4(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "=&v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3) : "vcc");
}
__device__ __forceinline__ void mac96f_5(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "=&v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4) : "vcc");
}
__device__ __forceinline__ void mac96f_6(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4, uint32_t x5, uint32_t y5) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %12, %13, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "=&v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5) : "vcc");
}
__device__ __forceinline__ void mac96f_7(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4, uint32_t x5, uint32_t y5, uint32_t x6, uint32_t y6) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %12, %13, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %14, %15, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "=&v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5), "v"(x6), "v"(y6) : "vcc");
}
__device__ __forceinline__ void mac96f_8(uint64_t &lo, uint32_t &c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4, uint32_t x5, uint32_t y5, uint32_t x6, uint32_t y6, uint32_t x7, uint32_t y7) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %12, %13, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %14, %15, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %16, %17, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "=&v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5), "v"(x6), "v"(y6), "v"(x7), "v"(y7) : "vcc");
}

// Montgomery product a*b/2^256 mod p by product scanning (FIPS): columns k = 0..14 of a*b + m*p are
// summed in a 96-bit accumulator; m[k] = column_k * (-p^-1) mod 2^32 zeroes the low word of each of the
// first eight columns.  136 multiply-adds, 2 VALU instructions each.
#endif  // __HIP_DEVICE_COMPILE__

#if !defined(__HIP_DEVICE_COMPILE__)
// Host build of the same product (proof assembly, key build, CPU unit tests): 4 x 64-bit limbs, CIOS with
// unsigned __int128 -- about 3x faster on x86-64 than the 32-bit product scanning below.  Same result bit for bit.
template <class PM>
inline Fp<PM> mul_host64(const Fp<PM> &a, const Fp<PM> &b) {
  typedef unsigned __int128 u128;
  uint64_t x[4], y[4], p[4], t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    x[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
    y[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
    p[i] = (uint64_t)PM::P[2 * i] | ((uint64_t)PM::P[2 * i + 1] << 32);
  }
  // -p^-1 mod 2^64 from the 32-bit constant by one Newton step: inv64 = inv32 * (2 + p0 * inv32)
  const uint64_t inv32 = PM::INV;
  const uint64_t inv64 = inv32 * (2 + p[0] * inv32);
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)x[j] * y[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * inv64;
    c = (u128)m * p[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  Fp<PM> r;
  for (int i = 0; i < 4; i++) { r.v[2 * i] = (uint32_t)t[i]; r.v[2 * i + 1] = (uint32_t)(t[i] >> 32); }
  return reduce_once(r);  // t < 2p < 2^255, t[4] == 0
}
#endif

template <class PM>
ZKR_HD Fp<PM> mul(const Fp<PM> &a, const Fp<PM> &b) {
#if !defined(__HIP_DEVICE_COMPILE__)
  return mul_host64(a, b);
#else
  const uint32_t *x = a.v, *y = b.v;
  constexpr const uint32_t *P = PM::P;
  uint32_t m[8], t[8];
  uint64_t lo = 0;
  uint32_t c2 = 0;
#define ZKR_NEXT_COL lo = (lo >> 32) | ((uint64_t)c2 << 32);
#define ZKR_FIN_LOW(K) m[K] = (uint32_t)lo * PM::INV; mac96c_1<P[0]>(lo, c2, m[K]); ZKR_NEXT_COL
  mac96f_1(lo, c2, x[0], y[0]);
  ZKR_FIN_LOW(0)
  mac96f_2(lo, c2, x[0], y[1], x[1], y[0]);
  mac96c_1<P[1]>(lo, c2, m[0]);
  ZKR_FIN_LOW(1)
  mac96f_3(lo, c2, x[0], y[2], x[1], y[1], x[2], y[0]);
  mac96c_2<P[2], P[1]>(lo, c2, m[0], m[1]);
  ZKR_FIN_LOW(2)
  mac96f_4(lo, c2, x[0], y[3], x[1], y[2], x[2], y[1], x[3], y[0]);
  mac96c_3<P[3], P[2], P[1]>(lo, c2, m[0], m[1], m[2]);
  ZKR_FIN_LOW(3)
  mac96f_5(lo, c2, x[0], y[4], x[1], y[3], x[2], y[2], x[3], y[1], x[4], y[0]);
  mac96c_4<P[4], P[3], P[2], P[1]>(lo, c2, m[0], m[1], m[2], m[3]);
  ZKR_FIN_LOW(4)
  mac96f_6(lo, c2, x[0], y[5], x[1], y[4], x[2], y[3], x[3], y[2], x[4], y[1], x[5], y[0]);
  mac96c_5<P[5], P[4], P[3], P[2], P[1]>(lo, c2, m[0], m[1], m[2], m[3], m[4]);
  ZKR_FIN_LOW(5)
  mac96f_7(lo, c2, x[0], y[6], x[1], y[5], x[2], y[4], x[3], y[3], x[4], y[2], x[5], y[1], x[6], y[0]);
  mac96c_6<P[6], P[5], P[4], P[3], P[2], P[1]>(lo, c2, m[0], m[1], m[2], m[3], m[4], m[5]);
  ZKR_FIN_LOW(6)
  mac96f_8(lo, c2, x[0], y[7], x[1], y[6], x[2], y[5], x[3], y[4], x[4], y[3], x[5], y[2], x[6], y[1], x[7], y[0]);
  mac96c_7<P[7], P[6], P[5], P[4], P[3], P[2], P[1]>(lo, c2, m[0], m[1], m[2], m[3], m[4], m[5], m[6]);
  ZKR_FIN_LOW(7)
  mac96f_7(lo, c2, x[1], y[7], x[2], y[6], x[3], y[5], x[4], y[4], x[5], y[3], x[6], y[2], x[7], y[1]);
  mac96c_7<P[7], P[6], P[5], P[4], P[3], P[2], P[1]>(lo, c2, m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
  t[0] = (uint32_t)lo; ZKR_NEXT_COL
  mac96f_6(lo, c2, x[2], y[7], x[3], y[6], x[4], y[5], x[5], y[4], x[6], y[3], x[7], y[2]);
  mac96c_6<P[7], P[6], P[5], P[4], P[3], P[2]>(lo, c2, m[2], m[3], m[4], m[5], m[6], m[7]);
  t[1] = (uint32_t)lo; ZKR_NEXT_COL
  mac96f_5(lo, c2, x[3], y[7], x[4], y[6], x[5], y[5], x[6], y[4], x[7], y[3]);
  mac96c_5<P[7], P[6], P[5], P[4], P[3]>(lo, c2, m[3], m[4], m[5], m[6], m[7]);
  t[2] = (uint32_t)lo; ZKR_NEXT_COL
  mac96f_4(lo, c2, x[4], y[7], x[5], y[6], x[6], y[5], x[7], y[4]);
  mac96c_4<P[7], P[6], P[5], P[4]>(lo, c2, m[4], m[5], m[6], m[7]);
  t[3] = (uint32_t)lo; ZKR_NEXT_COL
  mac96f_3(lo, c2, x[5], y[7], x[6], y[6], x[7], y[5]);
  mac96c_3<P[7], P[6], P[5]>(lo, c2, m[5], m[6], m[7]);
  t[4] = (uint32_t)lo; ZKR_NEXT_COL
  mac96f_2(lo, c2, x[6], y[7], x[7], y[6]);
  mac96c_2<P[7], P[6]>(lo, c2, m[6], m[7]);
  t[5] = (uint32_t)lo; ZKR_NEXT_COL
  mac96f_1(lo, c2, x[7], y[7]);
  mac96c_1<P[7]>(lo, c2, m[7]);
  t[6] = (uint32_t)lo; ZKR_NEXT_COL
  t[7] = (uint32_t)lo;  // column 15 is empty and (a*b + m*p)/2^256 < 2p < 2^255
#undef ZKR_NEXT_COL
#undef ZKR_FIN_LOW
  Fp<PM> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return reduce_once(r);
#endif
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
ZKR_HD Fp<PM> inv_inline(const Fp<PM> &a) {
  Fp<PM> r = Fp<PM>::one(), b = a;
  // the exponent's words come from the modulus table as they are needed: a local copy indexed by the loop counter would live in
  // scratch memory on the device (the 144 / 48 B per lane fixed_base_kernel had)
  for (int w = 0; w < 8; w++) {
    uint32_t e = PM::P[w] - (w == 0 ? 2u : 0u);  // low limb of both moduli is > 2
    const int bits = w == 7 ? 254 - 224 : 32;
    for (int i = 0; i < bits; i++) {
      if ((e >> i) & 1) r = mul(r, b);
      b = sqr(b);
    }
  }
  return r;
}
template <class PM>
ZKR_HD_COLD Fp<PM> inv(const Fp<PM> &a) { return inv_inline(a); }  // out of line: the host and the cold device paths

#include "field_fused.hpp"  // mul_sum2 / mul_sum4 (generated by tools/gen_mul_sum.py)

// a*b - c*d with one reduction
template <class PM>
ZKR_HD Fp<PM> mul_sub(const Fp<PM> &a, const Fp<PM> &b, const Fp<PM> &c, const Fp<PM> &d) {
  return mul_sum2(a, b, neg(c), d);
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
ZKR_HD Fq2 mul(const Fq2 &x, const Fq2 &y) {
#if defined(__HIP_DEVICE_COMPILE__)
  // schoolbook with one reduction per component: 2 x (128 + 72) multiply-adds and one negation, against
  // 3 x 136 plus five additions for Karatsuba
  return Fq2{mul_sum2(x.a, y.a, neg(x.b), y.b), mul_sum2(x.a, y.b, x.b, y.a)};
#else
  Fq t0 = mul(x.a, y.a), t1 = mul(x.b, y.b);  // Karatsuba, 3 Fq products
  Fq m = mul(add(x.a, x.b), add(y.a, y.b));
  return Fq2{sub(t0, t1), sub(sub(m, t0), t1)};
#endif
}
// x*y - z*w: each component is a sum of four products with one reduction
ZKR_HD Fq2 mul_sub(const Fq2 &x, const Fq2 &y, const Fq2 &z, const Fq2 &w) {
  Fq nza = neg(z.a);
  return Fq2{mul_sum4(x.a, y.a, neg(x.b), y.b, nza, w.a, z.b, w.b), mul_sum4(x.a, y.b, x.b, y.a, nza, w.b, neg(z.b), w.a)};
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
