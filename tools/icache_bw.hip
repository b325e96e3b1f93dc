// Does a loop body larger than the 64 KB instruction cache (shared by two CUs) slow a multiply-bound kernel down?
// Straight-line runs of v_mad_u64_u32 (8 bytes each, four independent accumulators), body sizes 16 KB .. 112 KB (a backward branch reaches 128 KB), two
// wavefronts per SIMD like the accumulation kernels.  Prints multiply-adds per second for every body size.
//   hipcc -O3 --offload-arch=gfx950 tools/icache_bw.hip -o tools/bin/icache_bw && tools/bin/icache_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int QUADS>
__global__ __launch_bounds__(256) void body_kernel(uint64_t *out, uint32_t x, uint32_t y, int iters) {
  uint64_t a = threadIdx.x, b = blockIdx.x, c = 3, d = 5;
  for (int i = 0; i < iters; i++) {
    asm volatile(".rept %6\n"
                 "v_mad_u64_u32 %0, vcc, %4, %5, %0\n"
                 "v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                 "v_mad_u64_u32 %2, vcc, %4, %5, %2\n"
                 "v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                 ".endr\n"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y), "n"(QUADS) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int QUADS>
static void run(uint64_t *d_out, int blocks) {
  const long total_quads = 7L << 15;                 // per thread: 2^20 multiply-adds whatever the body size
  int iters = (int)(total_quads / QUADS);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  body_kernel<QUADS><<<blocks, 256>>>(d_out, 12345u, 6789u, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  body_kernel<QUADS><<<blocks, 256>>>(d_out, 12345u, 6789u, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  double mads = double(blocks) * 256 * iters * QUADS * 4;
  printf("body %4d KB  %8.3f ms  %.3e multiply-adds/s\n", QUADS * 32 / 1024, ms, mads / (ms * 1e-3));
}

int main(int argc, char **argv) {
  int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
  int blocks = 256 * waves_per_simd;                 // 256 CUs x 4 SIMDs x waves / 4 wavefronts per block
  uint64_t *d_out; hipMalloc(&d_out, size_t(blocks) * 256 * 8);
  printf("%d wavefront(s) per SIMD\n", waves_per_simd);
  run<512>(d_out, blocks); run<1024>(d_out, blocks); run<1536>(d_out, blocks); run<1792>(d_out, blocks); run<2048>(d_out, blocks);
  run<2560>(d_out, blocks); run<3072>(d_out, blocks); run<3584>(d_out, blocks);
  return 0;
}
