// dep_chain.hip -- issue interval of DEPENDENT v_mad_u64_u32 (the accumulator chain of one column of the Montgomery product,
// field29_asm.hpp) against the number of independent chains a wavefront interleaves (C = 1, 2, 4) and the wavefronts a SIMD
// holds (W).  The accumulation kernels hold 2-3 wavefronts per SIMD: if one chain per wavefront cannot fill the multiplier at
// that occupancy, the product has to interleave columns.
// Build: hipcc -O3 --offload-arch=gfx950 tools/dep_chain.hip -o /tmp/dep_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int C>
__global__ __launch_bounds__(64) void k(uint64_t *out, int iters) {
  uint64_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
  uint32_t x = threadIdx.x * 2654435761u + 1, y = x ^ 0x9e3779b9u;
  for (int i = 0; i < iters; i++) {
    if (C == 1) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y) : "vcc");) }
    if (C == 2) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(a0), "+v"(a1) : "v"(x), "v"(y) : "vcc");) }
    if (C == 4) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + x;
}

template <int C>
static void run(const char *name, int W) {
  const int iters = 2000;
  int blocks = 1024 * W;  // one wavefront per block, W per SIMD
  uint64_t *d;
  hipMalloc(&d, (size_t)blocks * 64 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<C><<<blocks, 64>>>(d, iters);
  k<C><<<blocks, 64>>>(d, iters);
  hipEventRecord(e0);
  k<C><<<blocks, 64>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double insts = (double)iters * 64.0 * 4;
  printf("%-34s W=%d  %8.3f ms  %6.3f ns per wave64 instruction per SIMD  (%5.2f cycles at 2.35 GHz)\n", name, W, ms,
         ms * 1e6 / (insts * W), ms * 1e6 / (insts * W) * 2.35);
  hipFree(d);
}

int main() {
  for (int W : {1, 2, 3, 4, 8}) {
    run<1>("mad, one dependent chain", W);
    run<2>("mad, two chains interleaved", W);
    run<4>("mad, four chains interleaved", W);
  }
  return 0;
}
