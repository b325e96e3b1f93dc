// vcc_chain_test.hip -- does a carry chain of full-rate VALU adds (v_add_co / v_addc_co ... back to back, no wait
// states in between) produce correct 256-bit sums on gfx950?  The compiler pads its own chains with s_nop 1 after every
// instruction (VALU-writes-VCC -> VALU-reads-VCC wait states); this checks whether an inline-asm chain without the padding
// is safe, against the compiler's padded chain, on 2^22 threads x 4096 dependent iterations with carry-heavy operands.
//   hipcc -O3 --offload-arch=gfx950 tools/vcc_chain_test.hip -o /tmp/vcc_chain_test && /tmp/vcc_chain_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct U256 { uint32_t v[8]; };

__device__ __forceinline__ U256 add_builtin(const U256 &a, const U256 &b) {
  U256 s; uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
  return s;
}
__device__ __forceinline__ U256 sub_builtin(const U256 &a, const U256 &b) {
  U256 s; uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s.v[i] = __builtin_subc(a.v[i], b.v[i], c, &c);
  return s;
}
__device__ __forceinline__ U256 add_asm(const U256 &a, const U256 &b) {
  U256 s;
  asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
      "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %15, %23, vcc"
      : "=&v"(s.v[0]), "=&v"(s.v[1]), "=&v"(s.v[2]), "=&v"(s.v[3]), "=&v"(s.v[4]), "=&v"(s.v[5]), "=&v"(s.v[6]), "=&v"(s.v[7])
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
        "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  return s;
}
__device__ __forceinline__ U256 sub_asm(const U256 &a, const U256 &b) {
  U256 s;
  asm("v_sub_co_u32 %0, vcc, %8, %16\n\t"
      "v_subb_co_u32 %1, vcc, %9, %17, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %15, %23, vcc"
      : "=&v"(s.v[0]), "=&v"(s.v[1]), "=&v"(s.v[2]), "=&v"(s.v[3]), "=&v"(s.v[4]), "=&v"(s.v[5]), "=&v"(s.v[6]), "=&v"(s.v[7])
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
        "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  return s;
}

template <bool ASM>
__global__ void chain(U256 *out, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  U256 a, b;
  for (int i = 0; i < 8; i++) {
    uint32_t h = (t + 1) * 2654435761u + i * 40503u;
    a.v[i] = (h & 3) == 0 ? 0xFFFFFFFFu : h;          // carry-heavy: many all-ones limbs
    b.v[i] = (h >> 3 & 3) == 0 ? 1u : ~h + (i == 0);
  }
  for (int it = 0; it < iters; it++) {
    U256 s = ASM ? add_asm(a, b) : add_builtin(a, b);
    U256 d = ASM ? sub_asm(s, a) : sub_builtin(s, a);   // == b unless a carry was dropped
    for (int i = 0; i < 8; i++) { a.v[i] = s.v[i] ^ (d.v[i] >> 7); b.v[i] = d.v[i] + (s.v[(i + 1) & 7] | 1u); }
  }
  U256 r;
  for (int i = 0; i < 8; i++) r.v[i] = a.v[i] ^ b.v[i];
  out[t] = r;
}

int main() {
  const int n = 1 << 22, iters = 4096;
  U256 *d0, *d1;
  hipMalloc(&d0, (size_t)n * 32);
  hipMalloc(&d1, (size_t)n * 32);
  hipEvent_t e0, e1, e2;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  chain<false><<<n / 256, 256>>>(d0, 16);
  chain<true><<<n / 256, 256>>>(d1, 16);
  hipEventRecord(e0);
  chain<false><<<n / 256, 256>>>(d0, iters);
  hipEventRecord(e1);
  chain<true><<<n / 256, 256>>>(d1, iters);
  hipEventRecord(e2);
  hipDeviceSynchronize();
  float t0, t1;
  hipEventElapsedTime(&t0, e0, e1);
  hipEventElapsedTime(&t1, e1, e2);
  U256 *h0 = (U256 *)malloc((size_t)n * 32), *h1 = (U256 *)malloc((size_t)n * 32);
  hipMemcpy(h0, d0, (size_t)n * 32, hipMemcpyDeviceToHost);
  hipMemcpy(h1, d1, (size_t)n * 32, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (int i = 0; i < n; i++) bad += memcmp(&h0[i], &h1[i], 32) != 0;
  printf("compiler chain %.2f ms, asm chain without wait states %.2f ms, %zu of %d threads differ\n", t0, t1, bad, n);
  return bad != 0;
}
