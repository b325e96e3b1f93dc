// mul_ilp_test.hip -- how much of the Fq multiplier's rate depends on independent work per lane when only a few wavefronts
// share a SIMD (the accumulation kernels run at 2 per SIMD)?  CHAINS independent dependent-product chains per lane, the
// kernel compiled for WAVES wavefronts per SIMD and launched to exactly that occupancy.
//   hipcc -O3 --offload-arch=gfx950 -Isimple-zk-rollups_amd/csrc tools/mul_ilp_test.hip -o /tmp/mul_ilp_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "field.hpp"
using namespace zkr;

template <int CHAINS, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k(Fq *io, int iters) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x[CHAINS], y = io[i];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) { x[c] = y; x[c].v[c & 7] ^= (c + 1); }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = mul(x[c], y);
  }
  Fq s = x[0];
#pragma unroll
  for (int c = 1; c < CHAINS; c++) s = add(s, x[c]);
  io[i] = s;
}

template <int CHAINS, int WAVES>
void run(Fq *d) {
  const int blocks = 256 * 4 * WAVES / 4, iters = 2048;  // 256 CUs x 4 SIMDs x WAVES wavefronts, 4 wavefronts per block
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<CHAINS, WAVES><<<blocks, 256>>>(d, 16);
  hipEventRecord(e0);
  k<CHAINS, WAVES><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("waves/SIMD %d, chains/lane %d: %.1f G Fq-mul/s\n", WAVES, CHAINS, (double)blocks * 256 * iters * CHAINS / (ms * 1e-3) / 1e9);
}

int main() {
  Fq *d;
  hipMalloc(&d, (size_t)256 * 4 * 8 * 64 * 32);
  hipMemset(d, 0x11, (size_t)256 * 4 * 8 * 64 * 32);
  run<1, 1>(d); run<2, 1>(d); run<4, 1>(d);
  run<1, 2>(d); run<2, 2>(d); run<4, 2>(d);
  run<1, 4>(d); run<2, 4>(d); run<4, 4>(d);
  run<1, 8>(d);
  return 0;
}
