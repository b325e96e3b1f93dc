// valu_clock.hip -- what the chip itself says about the numbers the VALU roofline rests on (VERDICT r1 "weak" 3):
//   * the shader clock under an integer multiply load, read on the device: s_memtime (core-clock counter, clock64()) against
//     s_memrealtime (constant reference clock, wall_clock64()), over kernels that run for tens of milliseconds;
//   * cycles per wave64 instruction, in those core-clock cycles, with every SIMD holding W wavefronts: v_mad_u64_u32,
//     the v_mad_u64_u32 + v_addc_co_u32 pair of the Montgomery product's inner step, v_add_u32, v_mul_lo_u32, v_fma_f64.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_clock.hip -o /tmp/valu_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint64_t *ticks, int iters) {
  uint64_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
  uint32_t x = threadIdx.x * 2654435761u + 1, y = x ^ 0x9e3779b9u, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  double d0 = 1.0 + threadIdx.x, d1 = 2.0, d2 = 3.0, d3 = 4.0, e = 1.000001, f = 0.999999;
  uint64_t t0 = clock64(), r0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");) }
    else if (OP == 1) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_addc_co_u32 %4, vcc, 0, %4, vcc\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_addc_co_u32 %6, vcc, 0, %6, vcc\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_addc_co_u32 %7, vcc, 0, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y) : "vcc");) }
    else if (OP == 2) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));) }
    else if (OP == 3) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));) }
    else if (OP == 4) { REP64(asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e), "v"(f));) }
    else if (OP == 5) { REP64(asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "vcc");) }
    else if (OP == 7) { REP64(asm volatile("v_lshrrev_b64 %0, 29, %0\n v_lshrrev_b64 %1, 29, %1\n v_lshrrev_b64 %2, 29, %2\n v_lshrrev_b64 %3, 29, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    else if (OP == 8) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    else if (OP == 9) { REP64(asm volatile("v_alignbit_b32 %0, %1, %0, 29\n v_alignbit_b32 %1, %2, %1, 29\n v_alignbit_b32 %2, %3, %2, 29\n v_alignbit_b32 %3, %0, %3, 29" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));) }
    else if (OP == 10) { REP64(asm volatile("v_and_b32 %0, %4, %0\n v_and_b32 %1, %4, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));) }
    else if (OP == 6) { REP64(asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[22:23], %8, %9, %1\n v_addc_co_u32_e64 %4, s[20:21], 0, %4, s[20:21]\n v_addc_co_u32_e64 %5, s[22:23], 0, %5, s[22:23]\n v_mad_u64_u32 %2, s[20:21], %8, %9, %2\n v_mad_u64_u32 %3, s[22:23], %8, %9, %3\n v_addc_co_u32_e64 %6, s[20:21], 0, %6, s[20:21]\n v_addc_co_u32_e64 %7, s[22:23], 0, %7, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y) : "s20", "s21", "s22", "s23");) }
  }
  uint64_t t1 = clock64(), r1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + c0 + c1 + c2 + c3 + (uint64_t)(d0 + d1 + d2 + d3);
  if ((threadIdx.x & 63) == 0) {
    uint32_t w = blockIdx.x * 4 + threadIdx.x / 64;
    ticks[2 * w] = t1 - t0;
    ticks[2 * w + 1] = r1 - r0;
  }
}

template <int OP>
static void run(const char *name, int insts_per_rep, int W, double ref_khz) {
  const int iters = 6000;  // ~1.5 M instructions per wavefront: tens of milliseconds per launch, the clock has settled
  int blocks = 256 * W, waves = blocks * 4;
  uint64_t *d, *t;
  hipMalloc(&d, (size_t)blocks * 256 * 8);
  hipMalloc(&t, (size_t)waves * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(d, t, iters);  // warm: brings the clock up under this very load
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(d, t, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(2 * (size_t)waves);
  hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
  double clk = 0, ref = 0;
  for (int w = 0; w < waves; w++) { clk += (double)h[2 * w]; ref += (double)h[2 * w + 1]; }
  clk /= waves; ref /= waves;
  double insts = (double)iters * 64.0 * insts_per_rep;         // per wavefront
  double sclk_mhz = clk / ref * ref_khz / 1e3;
  printf("%-40s W=%d  %8.3f ms  sclk %7.1f MHz (device counters)  %6.3f core cycles per wave64 instruction  [host clock: %6.3f at that sclk]\n",
         name, W, ms, sclk_mhz, clk / (insts * W), ms * 1e-3 * sclk_mhz * 1e6 / (insts * W));
  hipFree(d); hipFree(t);
}

int main() {
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  int ckhz = 0;
  hipDeviceGetAttribute(&ckhz, hipDeviceAttributeClockRate, 0);
  printf("wall clock rate %d kHz, nominal core clock %d kHz\n", khz, ckhz);
  for (int W : {2, 8}) {
    run<0>("v_mad_u64_u32", 4, W, khz);
    run<1>("v_mad_u64_u32 + v_addc_co_u32 (per instr)", 8, W, khz);
    run<6>("same, two carry chains in SGPR pairs", 8, W, khz);
    run<5>("v_addc_co_u32 (vcc chain)", 4, W, khz);
    run<2>("v_add_u32", 4, W, khz);
    run<3>("v_mul_lo_u32", 4, W, khz);
    run<4>("v_fma_f64", 4, W, khz);
    run<7>("v_lshrrev_b64", 4, W, khz);
    run<8>("v_lshl_add_u64", 4, W, khz);
    run<9>("v_alignbit_b32", 4, W, khz);
    run<10>("v_and_b32 / v_lshrrev_b32", 4, W, khz);
  }
  return 0;
}
