// Random 64-byte (and 128-byte) gathers from a table in HBM, as the bucket accumulation issues them: every thread walks a
// list of random indices, one 64 B load per step (four dwordx4... here two uint4 pairs), nothing else.  Prints GB/s for table
// sizes from 16 MB (L2) over 128 MB (Infinity Cache) to 1 GB+ (HBM), and for a sequential walk as reference.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/gather_bw tools/gather_bw.hip && ./tools/bin/gather_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
template <int BYTES>
__global__ __launch_bounds__(256) void gather(const uint4 *tbl, const uint32_t *idx, uint32_t per_thread, uint32_t nthreads, uint4 *out) {
  extern __shared__ uint32_t pad[];  // occupancy limiter: 80 KB per workgroup = two workgroups per CU = two wavefronts per SIMD
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nthreads) return;
  if (per_thread == 0xffffffffu) pad[threadIdx.x] = t;
  uint4 acc = make_uint4(0, 0, 0, 0);
  uint32_t e = idx[(size_t)t * per_thread];
  for (uint32_t j = 0; j < per_thread; j++) {
    uint32_t en = j + 1 < per_thread ? idx[(size_t)t * per_thread + j + 1] : 0;
    const uint4 *p = tbl + (size_t)e * (BYTES / 16);
#pragma unroll
    for (int k = 0; k < BYTES / 16; k++) { uint4 v = p[k]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    e = en;
  }
  out[t] = acc;
}
// gather_bw <bytes> <table MB> <seq 0|1> <lds KB>: that one configuration only (four launches) -- what a counter pass wants:
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/bin/gather_bw 64 832 0 80
// requests 2^19 x 26 gathers of `bytes` (+ 4 B of index each): the known byte count FETCH_SIZE is calibrated against for THIS access
// pattern (profiles/r6_*_fetch_size_calibration.md)
int main(int argc, char **argv) {
  const uint32_t nthreads = 1u << 19, per = 26;
  const bool one = argc == 5;
  const int one_bytes = one ? atoi(argv[1]) : 0, one_seq = one ? atoi(argv[3]) : 0;
  const size_t one_mb = one ? (size_t)atoi(argv[2]) : 0, one_lds = one ? (size_t)atoi(argv[4]) * 1024 : 0;
  uint4 *out; hipMalloc(&out, (size_t)nthreads * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t lds : {(size_t)0, (size_t)80 * 1024}) for (int bytes : {64, 128}) for (size_t mb : {(size_t)64, (size_t)832, (size_t)17408, (size_t)69632}) for (int seq = 0; seq < 2; seq++) {
    if (!one && mb > 832) continue;  // the large tables (a 256-level G1 table of 2^20 points: 17 GB; four of them) only on request
    if (one && (bytes != one_bytes || mb != one_mb || seq != one_seq || lds != one_lds)) continue;
    size_t n = mb * 1024 * 1024 / bytes;
    uint4 *tbl; hipMalloc(&tbl, n * bytes); if (hipMemset(tbl, 1, n * bytes) != hipSuccess) { printf("table of %zu MB: allocation failed\n", mb); return 1; }
    std::vector<uint32_t> idx((size_t)nthreads * per);
    uint64_t st = 88172645463325252ull;
    for (size_t i = 0; i < idx.size(); i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; idx[i] = seq ? (uint32_t)(i % n) : (uint32_t)((st >> 11) % n); }
    uint32_t *d_idx; hipMalloc(&d_idx, idx.size() * 4); hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0);
      if (bytes == 64) gather<64><<<nthreads / 256, 256, lds>>>(tbl, d_idx, per, nthreads, out);
      else gather<128><<<nthreads / 256, 256, lds>>>(tbl, d_idx, per, nthreads, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%s %3d-byte %s gathers, table %5zu MB: %7.1f us for %.0f MB -> %6.0f GB/s\n", lds ? "2 waves/SIMD " : "full occupancy", bytes, seq ? "sequential" : "random    ", mb, best * 1e3,
           (double)nthreads * per * bytes / 1e6, (double)nthreads * per * bytes / (best * 1e-3) / 1e9);
    hipFree(tbl); hipFree(d_idx);
  }
  return 0;
}
