// kernels_ntt.hpp -- BN254 Fr NTT passes, QAP row evaluation and the element-wise stages of calcH.
//
// Path: websnark groth16GenProof's calcH (called at /root/reference/operator/src/snarks/common.ts:29;
// algorithm SURVEY.md App. B steps 1-3).  Vectors stay in STANDARD form in HBM, every constant
// (twiddles, QAP coefficients) is in Montgomery form, so montmul(data, const) needs no conversions.
//
// NTT decomposition (prototyped index-for-index in tests/test_ntt_plan.py):
//   radix-2 network split into passes; one pass = all stages with span 2^j, j in [lo,hi), executed
//   inside LDS on a tile of 2^(hi-lo) rows x W columns (W consecutive elements -> 32*W contiguous
//   bytes per row in HBM).  DIF passes run top-down (natural in -> bit-reversed out), DIT passes
//   bottom-up (bit-reversed in -> natural out), so no explicit permutation is ever done.
//   Inside a pass the stages run in pairs on four elements held in registers (one barrier per two stages).
//   LDS tile is limb-major (SoA): lane l touches bank (l mod 32) for every ds_read_b32 -> conflict free.
//   Twiddles w_{2s}^x come from one table per key (w_{2m}^k, k<m, serves all spans up to m and the
//   coset factors) plus a 32 KB table w_2048^k shared by every contiguous pass (L1/L2 resident).
#pragma once
#include "curve.hpp"

namespace zkr {

// Threads of an NTT workgroup (template parameter of the pass kernel), chosen per transform size by run_ntt (zkr_prove.hip).
// 512 (two butterflies per thread and stage) is the fastest pass in isolation (256 / 1024: 1.13 / 1.07 against 1.04 ms per
// calcH at 2^20), but its two wavefronts per SIMD need 2 x 128 VGPRs and do not fit beside two accumulation wavefronts
// (2 x 176 of 512): under a running accumulation the workgroup waits for a CU to drain.  A 256-thread workgroup (one
// wavefront per SIMD, 128 VGPRs) co-resides.  Same-box rounds: at 2^20 512 wins (136.8 against 135.0 proofs/s: accumulation
// wavefronts are short-lived there); from 2^21 on, where a bucket's chain is longer and the preparation chain paces the
// pipeline, 256 wins (2^21: 71.5 against 68.0; 2^22: 36.5 against 34.9; 2^24: 9.06 against 8.47 proofs/s).
constexpr int NTT_THREADS_SMALL = 512, NTT_THREADS_LARGE = 256;
constexpr int NTT_LARGE_LOG = 21;  // transforms of 2^21 and more take the 256-thread workgroups
constexpr int NTT_TILE_LOG = 11;   // 2048 elements = 64 KB of LDS per workgroup (2 workgroups / CU)
constexpr int NTT_STRIDED_LOG = 9; // stages per strided pass
constexpr int NTT_W_LOG = 2;       // 4 columns = 128 B contiguous per row in a strided pass
constexpr int TWL_LOG = 10;        // local table: w_2048^k, k < 1024

struct Vec32 { uint4 a, b; };

__device__ __forceinline__ Fr load_fr(const Fr *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ void store_fr(Fr *p, const Fr &r) {
  uint4 *q = reinterpret_cast<uint4 *>(p);
  q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
  q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// w_{2s}^x (inverse: w_{2s}^-x) from a table T[k] = w_{2^(tlog+1)}^k, k < 2^tlog; needs slog <= tlog, x < s
__device__ __forceinline__ Fr tw_lookup(const Fr *table, int tlog, int slog, uint32_t x, bool inverse) {
  uint32_t idx = x << (tlog - slog);
  if (!inverse) return load_fr(table + idx);
  if (idx == 0) return Fr::one();
  return neg(load_fr(table + ((1u << tlog) - idx)));  // w^(2^tlog) = -1
}

enum NttPre { PRE_NONE = 0, PRE_COSET = 1, PRE_MUL = 2 };

struct NttPassArgs {
  const Fr *in0;   // input (may alias out)
  const Fr *in1;   // second operand for PRE_MUL
  Fr *out;
  const Fr *tw;    // w_{2^(tlog+1)}^k table
  const Fr *twl;   // w_2048^k table
  int tlog;        // log2 of tw entries
  int L;           // transform size 2^L
  int lo, hi;      // stages [lo,hi)
  int wlog;        // log2 columns per tile (0 when lo == 0)
  int inverse;     // use inverse twiddles
  int pre;         // NttPre
  int prio;        // wave priority (s_setprio) of the pass: the preparation chain paces the pipeline once the accumulations are fast
};

// one pass; grid = 2^L / tile, block = NTT_THREADS, dynamic LDS = 32 * tile bytes
template <bool DIF, int NTT_THREADS>
static __global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(NttPassArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
  const int nb = a.hi - a.lo;
  const uint32_t rows = 1u << nb, W = 1u << a.wlog;
  const uint32_t tile = rows << a.wlog;
  const uint32_t lb = (1u << a.lo) >> a.wlog;  // tiles per q
  const uint32_t q = blockIdx.x / lb, l0 = (blockIdx.x % lb) << a.wlog;
  const size_t base = ((size_t)blockIdx.y << a.L) + ((size_t)q << a.hi) + l0;  // blockIdx.y: transform of a fused batch (vectors end to end)

  for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
    uint32_t r = e >> a.wlog, c = e & (W - 1);
    size_t gi = base + ((size_t)r << a.lo) + c;
    Fr x = load_fr(a.in0 + gi);
    if (a.pre == PRE_COSET) {
      uint32_t br = __brev((uint32_t)gi << (32 - a.L));  // coefficient index of a bit-reversed position (low L bits of gi: position within its transform)
      x = mul(x, load_fr(a.tw + ((size_t)br << (a.tlog - a.L))));
    } else if (a.pre == PRE_MUL) {
      x = mul(x, load_fr(a.in1 + gi));
    }
#pragma unroll
    for (int k = 0; k < 8; k++) lds[k * tile + e] = x.v[k];
  }
  __syncthreads();

  // Stages run in PAIRS on four elements held in registers (one LDS round trip and one barrier per two stages; index
  // math modelled in tests/test_ntt_plan.py `double`); an odd stage count starts with one single stage.
  auto twiddle = [&](int slog, uint32_t imods) {
    return slog <= TWL_LOG ? tw_lookup(a.twl, TWL_LOG, slog, imods, a.inverse != 0) : tw_lookup(a.tw, a.tlog, slog, imods, a.inverse != 0);
  };
  auto lds_get = [&](uint32_t e) {
    Fr x;
#pragma unroll
    for (int k = 0; k < 8; k++) x.v[k] = lds[k * tile + e];
    return x;
  };
  auto lds_put = [&](uint32_t e, const Fr &x) {
#pragma unroll
    for (int k = 0; k < 8; k++) lds[k * tile + e] = x.v[k];
  };
  auto butterfly = [&](Fr &u, Fr &v, const Fr &w) {
    if (DIF) {
      Fr d = sub(u, v);
      u = add(u, v);
      v = mul(d, w);
    } else {
      Fr t = mul(v, w);
      v = sub(u, t);
      u = add(u, t);
    }
  };
  auto single_stage = [&](int j) {
    const uint32_t sl = 1u << j;
    for (uint32_t b = threadIdx.x; b < (tile >> 1); b += NTT_THREADS) {
      uint32_t c = b & (W - 1), kk = b >> a.wlog;
      uint32_t r0 = ((kk >> j) << (j + 1)) | (kk & (sl - 1));
      uint32_t e0 = (r0 << a.wlog) + c, e1 = e0 + (sl << a.wlog);
      Fr w = twiddle(j + a.lo, ((kk & (sl - 1)) << a.lo) + l0 + c);
      Fr u = lds_get(e0), v = lds_get(e1);
      butterfly(u, v, w);
      lds_put(e0, u);
      lds_put(e1, v);
    }
    __syncthreads();
  };
  auto double_stage = [&](int j) {  // stages j and j + 1
    const uint32_t sl = 1u << j;
    for (uint32_t t = threadIdx.x; t < (tile >> 2); t += NTT_THREADS) {
      uint32_t c = t & (W - 1), kq = t >> a.wlog;
      uint32_t xlow = kq & (sl - 1);
      uint32_t r00 = ((kq >> j) << (j + 2)) | xlow;
      uint32_t e00 = (r00 << a.wlog) + c, e01 = e00 + (sl << a.wlog), e10 = e00 + (sl << (a.wlog + 1)), e11 = e10 + (sl << a.wlog);
      uint32_t im = (xlow << a.lo) + l0 + c;
      Fr wa = twiddle(j + a.lo, im), wb0 = twiddle(j + 1 + a.lo, im), wb1 = twiddle(j + 1 + a.lo, im + (sl << a.lo));
      Fr x00 = lds_get(e00), x01 = lds_get(e01), x10 = lds_get(e10), x11 = lds_get(e11);
      if (DIF) {  // stage j + 1 first
        butterfly(x00, x10, wb0);
        butterfly(x01, x11, wb1);
        butterfly(x00, x01, wa);
        butterfly(x10, x11, wa);
      } else {    // stage j first
        butterfly(x00, x01, wa);
        butterfly(x10, x11, wa);
        butterfly(x00, x10, wb0);
        butterfly(x01, x11, wb1);
      }
      lds_put(e00, x00);
      lds_put(e01, x01);
      lds_put(e10, x10);
      lds_put(e11, x11);
    }
    __syncthreads();
  };
  if (DIF) {
    int j = nb - 1;
    if (nb & 1) { single_stage(j); j--; }
    for (; j >= 1; j -= 2) double_stage(j - 1);
  } else {
    int j = 0;
    if (nb & 1) { single_stage(0); j = 1; }
    for (; j + 1 < nb; j += 2) double_stage(j);
  }

  for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
    uint32_t r = e >> a.wlog, c = e & (W - 1);
    size_t gi = base + ((size_t)r << a.lo) + c;
    Fr x;
#pragma unroll
    for (int k = 0; k < 8; k++) x.v[k] = lds[k * tile + e];
    store_fr(a.out + gi, x);
  }
}

// T[k] = g^k (Montgomery), k < n, g given in Montgomery form: thread k does square-and-multiply
static __global__ void twiddle_table_kernel(Fr *T, uint32_t n, Fr g) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  Fr r = Fr::one(), b = g;
  for (uint32_t e = k; e; e >>= 1) {
    if (e & 1) r = mul(r, b);
    b = sqr(b);
  }
  store_fr(T + k, r);
}

// witness ingest: reduce every 256-bit word below r (values from calculateWitness already are; this
// makes the path total for any buffer binarifyWitness can produce, binarify.ts:18-26)
static __global__ void ingest_kernel(const Fr *in, Fr *out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr x = load_fr(in + i);
#pragma unroll 1
  for (int k = 0; k < 5; k++) x = reduce_once(x);  // 2^256 < 6r
  store_fr(out + i, x);
}

// QAP evaluation on the domain: out[c] = sum_k coef[k] * w[col[k]] over CSR row c.
// coef Montgomery, w standard -> out standard.  One thread per constraint row: rows are 1-3 terms, except for the
// occasional wide row (64-term bit-packing rows, a handful per few thousand), which would leave 63 lanes of its
// wavefront waiting for one; rows wider than SPMV_WIDE are left to spmv_wide_kernel (one wavefront per row).
constexpr uint32_t SPMV_WIDE = 8;
static __global__ void spmv_kernel(const uint32_t *row_ptr, const uint32_t *col, const Fr *coef, const Fr *w, Fr *out, uint32_t m, uint32_t n) {
  uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m) return;
  w += (size_t)blockIdx.y * n;    // blockIdx.y: witness of a fused batch
  out += (size_t)blockIdx.y * m;
  uint32_t k0 = row_ptr[c], k1 = row_ptr[c + 1];
  if (k1 - k0 > SPMV_WIDE) return;
  Fr acc = Fr::zero();
  for (uint32_t k = k0; k < k1; k++) acc = add(acc, mul(load_fr(coef + k), load_fr(w + col[k])));
  store_fr(out + c, acc);
}
// wide[i] = index of the i-th row wider than SPMV_WIDE; one wavefront per row, terms strided over the lanes, LDS tree
static __global__ __launch_bounds__(64) void spmv_wide_kernel(const uint32_t *row_ptr, const uint32_t *col, const Fr *coef, const Fr *w, Fr *out,
                                                            const uint32_t *wide, uint32_t n_wide, uint32_t m, uint32_t n) {
  __shared__ uint32_t sh[8 * 64];
  if (blockIdx.x >= n_wide) return;
  w += (size_t)blockIdx.y * n;
  out += (size_t)blockIdx.y * m;
  const uint32_t c = wide[blockIdx.x], lane = threadIdx.x;
  uint32_t k0 = row_ptr[c], k1 = row_ptr[c + 1];
  Fr acc = Fr::zero();
  for (uint32_t k = k0 + lane; k < k1; k += 64) acc = add(acc, mul(load_fr(coef + k), load_fr(w + col[k])));
#pragma unroll
  for (int i = 0; i < 8; i++) sh[i * 64 + lane] = acc.v[i];
  __syncthreads();
  for (uint32_t s = 32; s > 0; s >>= 1) {
    if (lane < s) {
      Fr o;
#pragma unroll
      for (int i = 0; i < 8; i++) { acc.v[i] = sh[i * 64 + lane]; o.v[i] = sh[i * 64 + lane + s]; }
      acc = add(acc, o);
#pragma unroll
      for (int i = 0; i < 8; i++) sh[i * 64 + lane] = acc.v[i];
    }
    __syncthreads();
  }
  if (lane == 0) store_fr(out + c, acc);
}

// h (bit-reversed order) = C1*S' - C2 * g^-i * D'   (DESIGN.md "calcH on the GPU"); S', D' are the
// unscaled inverse-DIF outputs of a.b and A(gw^c).B(gw^c); i = bitrev(pos).
static __global__ void combine_h_kernel(const Fr *S, const Fr *D, Fr *h, const Fr *tw, int tlog, int L, Fr c1, Fr c2) {
  uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= (1u << L)) return;
  const size_t boff = (size_t)blockIdx.y << L;  // blockIdx.y: proof of a fused batch
  S += boff; D += boff; h += boff;
  uint32_t i = __brev(pos) >> (32 - L);
  Fr ginv = tw_lookup(tw, tlog, L, i, true);  // g^-i, g = w_{2m}
  Fr s = mul(load_fr(S + pos), c1);
  Fr d = mul(mul(load_fr(D + pos), ginv), c2);
  store_fr(h + pos, sub(s, d));
}

// out[i] = in[bitrev(i)]  (test hooks only)
static __global__ void bitrev_copy_kernel(const Fr *in, Fr *out, int L) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << L)) return;
  uint32_t j = L ? __brev(i) >> (32 - L) : 0;
  store_fr(out + i, load_fr(in + j));
}
// x[i] *= c
static __global__ void scale_kernel(Fr *x, size_t n, Fr c) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_fr(x + i, mul(load_fr(x + i), c));
}

}  // namespace zkr
