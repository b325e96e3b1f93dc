// valu_rates.hip -- issue-rate microbenchmark of the VALU instructions a 254-bit Montgomery multiplier can be
// built from on gfx950 (cycles per wave64 instruction with every SIMD saturated).  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o tools/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t *out, int iters) {
  uint64_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
  uint32_t x = threadIdx.x * 2654435761u + 1, y = x ^ 0x9e3779b9u;
  double d0 = 1.0 + threadIdx.x, d1 = 2.0 + threadIdx.x, d2 = 3.0, d3 = 4.0, e = 1.000001, f = 0.999999;
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) {  // v_mad_u64_u32, 4 independent chains
      REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");)
    } else if (OP == 1) {  // mad + addc pairs (the current multiplier's inner step)
      REP64(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_addc_co_u32 %4, vcc, 0, %4, vcc\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_addc_co_u32 %6, vcc, 0, %6, vcc\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_addc_co_u32 %7, vcc, 0, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y) : "vcc");)
    } else if (OP == 2) {  // v_fma_f64
      REP64(asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e), "v"(f));)
    } else if (OP == 3) {  // v_lshl_add_u64 (64-bit add)
      REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 4) {  // 64-bit add as v_add_co_u32 + v_addc_co_u32
      REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc\n v_add_co_u32 %2, vcc, %2, %0\n v_addc_co_u32 %3, vcc, %3, %1, vcc" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "vcc");)
    } else if (OP == 5) {  // v_mul_lo_u32
      REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));)
    } else if (OP == 6) {  // v_mul_hi_u32
      REP64(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));)
    } else if (OP == 7) {  // v_mad_u32_u24
      REP64(asm volatile("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_mad_u32_u24 %2, %4, %5, %2\n v_mad_u32_u24 %3, %4, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y));)
    } else if (OP == 8) {  // v_add_u32 (plain full-rate reference)
      REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));)
    } else if (OP == 9) {  // v_mul_f64
      REP64(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e));)
    } else if (OP == 10) {  // v_mad_u64_u32 single dependent chain
      REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y) : "vcc");)
    } else if (OP == 11) {  // v_fma_f64 single dependent chain
      REP64(asm volatile("v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0" : "+v"(d0) : "v"(e), "v"(f));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + c0 + c1 + c2 + c3 + (uint64_t)(d0 + d1 + d2 + d3);
}

template <int OP>
static void run(const char *name, int insts_per_rep, int waves_per_simd) {
  const int iters = 200;
  int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per 256-thread block = 1 per SIMD)
  uint64_t *d;
  hipMalloc(&d, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<OP><<<blocks, 256>>>(d, 2);
  hipEventRecord(e0);
  rate_kernel<OP><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double wave_insts = (double)blocks * 4 * iters * 64.0 * insts_per_rep;      // per-wave instruction count, all waves
  double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;                              // 1024 SIMDs at the nominal 2.4 GHz
  printf("%-44s waves/SIMD %d: %7.3f ms  %6.2f cycles per wave instruction\n", name, waves_per_simd, ms, simd_cycles / wave_insts);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_mad_u64_u32 (4 chains)", 4, w);
    run<10>("v_mad_u64_u32 (1 dependent chain)", 4, w);
    run<1>("v_mad_u64_u32 + v_addc_co_u32 (per pair)", 4, w);
    run<2>("v_fma_f64 (4 chains)", 4, w);
    run<11>("v_fma_f64 (1 dependent chain)", 4, w);
    run<9>("v_mul_f64", 4, w);
    run<3>("v_lshl_add_u64", 4, w);
    run<4>("v_add_co_u32 + v_addc_co_u32 (per pair)", 2, w);
    run<5>("v_mul_lo_u32", 4, w);
    run<6>("v_mul_hi_u32", 4, w);
    run<7>("v_mad_u32_u24", 4, w);
    run<8>("v_add_u32", 4, w);
  }
  return 0;
}
