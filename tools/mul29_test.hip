// mul29_test.hip -- throughput of variants of the 9 x 29-bit-limb Montgomery product (csrc/field29.hpp) on gfx950, all SIMDs
// at W wavefronts each, next to the 8 x 32-bit product of field.hpp.  The variants differ in how a column's carry moves on:
//   0: alignbit builtin + 32-bit shift      1: plain 64-bit shift (v_lshrrev_b64)      2: the two 32-bit shifts as inline assembly
// Build: hipcc -O3 --offload-arch=gfx950 -Isimple-zk-rollups_amd/csrc tools/mul29_test.hip -o /tmp/mul29_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "field29.hpp"
using namespace zkr;

template <int VAR>
__device__ __forceinline__ uint64_t shr29v(uint64_t a) {
  if (VAR == 1) return a >> 29;
  uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32), nlo, nhi;
  if (VAR == 0) { nlo = __builtin_amdgcn_alignbit(hi, lo, 29); nhi = hi >> 29; }
  else {  // 2: the two 32-bit shifts as opaque instructions (the optimiser folds the builtin form back into one 64-bit shift)
    asm("v_alignbit_b32 %0, %1, %2, 29" : "=v"(nlo) : "v"(hi), "v"(lo));
    asm("v_lshrrev_b32 %0, 29, %1" : "=v"(nhi) : "v"(hi));
  }
  return ((uint64_t)nhi << 32) | nlo;
}
template <int VAR>
__device__ __forceinline__ void mulv(uint32_t (&out)[9], const uint32_t (&a)[9], const uint32_t (&b)[9]) {
  using PM = Fq29;
  uint32_t m[9];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
#pragma unroll
    for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 0 && j <= 8) acc += (uint64_t)a[i] * b[j]; }
#pragma unroll
    for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 1 && j <= 8 && i < k) acc += (uint64_t)m[i] * PM::P[j]; }
    if (k < 9) { m[k] = ((uint32_t)acc * PM::INV) & M29; acc += (uint64_t)m[k] * PM::P[0]; acc = shr29v<VAR>(acc); }
    else { out[k - 9] = (uint32_t)acc & M29; acc = shr29v<VAR>(acc); }
  }
  out[8] = (uint32_t)acc;
}

// variant 3: every multiply-add an opaque instruction chained on ONE accumulator (no per-column 64-bit additions: the
// optimiser cannot re-associate the sums into partial chains); the price is a fully serial chain inside a product
__device__ __forceinline__ void madc(uint64_t &acc, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void madk(uint64_t &acc, uint32_t a, uint32_t k) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc");
}
__device__ __forceinline__ void mulv3(uint32_t (&out)[9], const uint32_t (&a)[9], const uint32_t (&b)[9]) {
  using PM = Fq29;
  uint32_t m[9];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
#pragma unroll
    for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 0 && j <= 8) madc(acc, a[i], b[j]); }
#pragma unroll
    for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 1 && j <= 8 && i < k) madk(acc, m[i], PM::P[j]); }
    if (k < 9) { m[k] = ((uint32_t)acc * PM::INV) & M29; madk(acc, m[k], PM::P[0]); acc >>= 29; }
    else { out[k - 9] = (uint32_t)acc & M29; acc >>= 29; }
  }
  out[8] = (uint32_t)acc;
}
template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k29v3(uint32_t *io, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a[9], b[9], c[9], d[9];
#pragma unroll
  for (int i = 0; i < 9; i++) { a[i] = io[t * 9 + i] & M29; b[i] = a[i] ^ 5; c[i] = a[i] ^ 9; d[i] = a[i] ^ 17; }
  for (int it = 0; it < iters; it++) { mulv3(a, a, b); mulv3(c, c, d); mulv3(b, b, a); mulv3(d, d, c); }   // two independent pairs
#pragma unroll
  for (int i = 0; i < 9; i++) io[t * 9 + i] = a[i] + b[i] + c[i] + d[i];
}

template <int VAR, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k29(uint32_t *io, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a[9], b[9], c[9], d[9];
#pragma unroll
  for (int i = 0; i < 9; i++) { a[i] = io[t * 9 + i] & M29; b[i] = a[i] ^ 5; c[i] = a[i] ^ 9; d[i] = a[i] ^ 17; }
  for (int it = 0; it < iters; it++) { mulv<VAR>(a, a, b); mulv<VAR>(b, b, c); mulv<VAR>(c, c, d); mulv<VAR>(d, d, a); }
#pragma unroll
  for (int i = 0; i < 9; i++) io[t * 9 + i] = a[i] + b[i] + c[i] + d[i];
}
template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k32(Fq *io, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = io[t], b = a, c = a, d = a;
  b.v[0] ^= 1; c.v[1] ^= 2; d.v[2] ^= 3;
  for (int it = 0; it < iters; it++) { a = mul(a, b); b = mul(b, c); c = mul(c, d); d = mul(d, a); }
  io[t] = add(add(a, b), add(c, d));
}
// the additions / subtractions of one mixed group addition next to its products: lazy 29-bit forms vs 32-bit carry chains
template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void kadd29(uint32_t *io, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  L29<Fq29, 3> a, b;
#pragma unroll
  for (int i = 0; i < 9; i++) { a.v[i] = io[t * 9 + i] & M29; b.v[i] = a.v[i] ^ 5; }
  for (int it = 0; it < iters; it++) {
    auto s = sub(a, b); auto u = add(s.to<8>(), b); auto w = sub(u.to<13>(), a);   // three carry sweeps
#pragma unroll
    for (int i = 0; i < 9; i++) { a.v[i] = w.v[i] & M29; b.v[i] ^= s.v[i] & 0xff; }
  }
#pragma unroll
  for (int i = 0; i < 9; i++) io[t * 9 + i] = a.v[i] + b.v[i];
}

template <class K>
static void run(const char *name, K kern, int waves, double per_iter, int iters) {
  int blocks = 256 * waves;
  void *d;
  hipMalloc(&d, (size_t)blocks * 256 * 40);
  hipMemset(d, 0x5a, (size_t)blocks * 256 * 40);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern(blocks, d, iters);
  hipEventRecord(e0);
  kern(blocks, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-46s W=%d  %8.3f ms  %7.1f G ops/s\n", name, waves, ms, (double)blocks * 256 * iters * per_iter / (ms * 1e-3) / 1e9);
  hipFree(d);
}
#define RUN29(VAR, W) run("mul 9x29, variant " #VAR, [](int b, void *d, int it) { k29<VAR, W><<<b, 256>>>((uint32_t *)d, it); }, W, 4, 4000)
#define RUN29V3(W) run("mul 9x29, variant 3 (opaque chained mads)", [](int b, void *d, int it) { k29v3<W><<<b, 256>>>((uint32_t *)d, it); }, W, 4, 4000)
#define RUN32(W) run("mul 8x32 (field.hpp)", [](int b, void *d, int it) { k32<W><<<b, 256>>>((Fq *)d, it); }, W, 4, 4000)
#define RUNADD(W) run("sub + add + sub on 9x29 (three carry sweeps)", [](int b, void *d, int it) { kadd29<W><<<b, 256>>>((uint32_t *)d, it); }, W, 3, 20000)
int main() {
  RUN32(2); RUN29(0, 2); RUN29(1, 2); RUN29(2, 2); RUN29V3(2); RUNADD(2);
  RUN32(4); RUN29(0, 4); RUN29(1, 4); RUN29(2, 4); RUN29V3(4); RUNADD(4);
  RUN32(8); RUN29(0, 8); RUN29(1, 8); RUN29(2, 8); RUN29V3(8);
  return 0;
}
