// kernels_msm.cuh -- multi-scalar multiplication over G1 / G2 (Pippenger bucket method) and the
// fixed-base kernels of the device-side setup.
//
// Path: the five multiexps of websnark's groth16GenProof (call site
// /root/reference/operator/src/snarks/common.ts:29).  Integer VALU work only (v_mad_u64_u32): no MFMA.
//
// Pipeline per MSM (all on one stream, no host round trip until the K window sums come back):
//   digits   : signed c-bit digits of every scalar, once per scalar vector (u16 codes, window-major)
//   hist     : workgroup per (window, chunk): bucket occupancies counted with LDS atomics
//   scan     : per-bucket prefix over chunks, then exclusive prefix sum -> bucket offsets; buckets larger
//              than `big_thresh` are listed
//   scatter  : (point index, sign) entries grouped by bucket, cursors in LDS (counting sort)
//   accum    : one thread per bucket, XYZZ += affine point (8M+2S), points gathered from the key table
//   big      : one workgroup per oversized bucket (0/1-heavy witnesses), LDS tree of XYZZ sums
//   reduce   : sum_b b*B_b per window: thread per group of g buckets (running sums) + small multiple
//   final    : one workgroup per window adds the group results -> K window sums (host does Horner)
#pragma once
#include "curve.cuh"

namespace zkr {

constexpr int MSM_THREADS = 256;
constexpr int MSM_MAX_WINDOWS = 64;
constexpr uint32_t BIG_CAP = 1024;  // oversized buckets tracked per MSM

template <class F> struct PointBytes;
template <> struct PointBytes<Fq> { static constexpr int N16 = 4; };   // 64 B affine
template <> struct PointBytes<Fq2> { static constexpr int N16 = 8; };  // 128 B affine

template <class T>
__device__ __forceinline__ T load_pod(const T *p) {
  static_assert(sizeof(T) % 16 == 0, "16-byte granules");
  T r;
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  uint4 *d = reinterpret_cast<uint4 *>(&r);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = q[i];
  return r;
}
template <class T>
__device__ __forceinline__ void store_pod(T *p, const T &v) {
  static_assert(sizeof(T) % 16 == 0, "16-byte granules");
  uint4 *q = reinterpret_cast<uint4 *>(p);
  const uint4 *s = reinterpret_cast<const uint4 *>(&v);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; i++) q[i] = s[i];
}

struct MsmGeom {
  uint32_t n;        // points
  int c;             // window bits
  int K;             // windows
  uint32_t nbw;      // buckets per window = 2^(c-1)
  uint32_t big_thresh;
  int glog;          // reduce group = 2^glog buckets
};

// Signed-digit recoding state: scalar kept in 8 registers and shifted right by c each window so the
// digit is always the low bits (no runtime-indexed register array -> no scratch).
struct DigitIter {
  uint32_t s[8];
  uint32_t carry;
  __device__ __forceinline__ void init(const uint32_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    carry = 0;
  }
  __device__ __forceinline__ bool is_zero() const {
    return (s[0] | s[1] | s[2] | s[3] | s[4] | s[5] | s[6] | s[7] | carry) == 0;
  }
  // returns signed digit in [-2^(c-1), 2^(c-1)]
  __device__ __forceinline__ int next(int c) {
    uint32_t raw = (s[0] & ((1u << c) - 1)) + carry;
#pragma unroll
    for (int i = 0; i < 7; i++) s[i] = (s[i] >> c) | (s[i + 1] << (32 - c));
    s[7] >>= c;
    int d = (int)raw;
    if (raw > (1u << (c - 1))) { d -= (1 << c); carry = 1; } else { carry = 0; }
    return d;
  }
};

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// ---------------------------------------------------------------- digit sort (LDS counting sort per window)
// Stage 1, once per SCALAR VECTOR (w serves the A, B1, B2 and C tables): signed c-bit digits of every
// scalar as u16 codes, window-major: dig[k * stride + i] = 0 (digit 0: no entry) or
// 1 + ((|d| - 1) << 1 | (d < 0)).  |d| <= 2^(c-1) <= 32768 and d > -2^(c-1), so the code fits 16 bits.
static __global__ __launch_bounds__(MSM_THREADS) void msm_digits_kernel(const Fr *scalars, uint32_t n, int c, int K, size_t stride, uint16_t *dig) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  DigitIter it;
  it.init(scalars[i].v);
  for (int k = 0; k < K; k++) {
    int d = it.next(c);
    uint32_t mag = (uint32_t)(d < 0 ? -d : d);
    dig[(size_t)k * stride + i] = d == 0 ? (uint16_t)0 : (uint16_t)(1u + (((mag - 1u) << 1) | (d < 0 ? 1u : 0u)));
  }
}

// Stages 2 and 4 run one workgroup per (window k, chunk j of the table): the window's 2^(c-1) bucket
// counters live in LDS (128 KB at c = 16, one workgroup per CU), so the 16 M increments of a 2^20-point
// MSM are LDS atomics instead of L2/fabric atomics.  blockIdx -> (k, j) keeps all chunks of a window on one
// XCD (block b runs on XCD b % 8; K is a multiple of 8 at full size) so their scattered 4-byte entry
// writes merge in one L2.
constexpr int SORT_THREADS = 1024;

// cnt[(k * J + j) * nbw + b] = occupancy of bucket b of window k within chunk j
static __global__ __launch_bounds__(SORT_THREADS) void msm_hist_kernel(const uint16_t *dig, size_t stride, const uint32_t *sidx, uint32_t n, int K, uint32_t nbw,
                                                                     uint32_t J, uint32_t chunk, uint32_t *cnt) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_bkt[];
  const uint32_t k = blockIdx.x % (uint32_t)K, j = blockIdx.x / (uint32_t)K;
  for (uint32_t b = threadIdx.x; b < nbw; b += SORT_THREADS) s_bkt[b] = 0;
  __syncthreads();
  const uint16_t *dk = dig + (size_t)k * stride;
  uint32_t i0 = j * chunk, i1 = min(i0 + chunk, n);
  for (uint32_t i = i0 + threadIdx.x; i < i1; i += SORT_THREADS) {
    uint32_t code = dk[sidx ? sidx[i] : i];
    if (code) atomicAdd(&s_bkt[(code - 1u) >> 1], 1u);
  }
  __syncthreads();
  uint32_t *out = cnt + ((size_t)k * J + j) * nbw;
  for (uint32_t b = threadIdx.x; b < nbw; b += SORT_THREADS) out[b] = s_bkt[b];
}

// per bucket: exclusive prefix over the J chunks (in place) and the bucket total -> counts[]
static __global__ __launch_bounds__(MSM_THREADS) void msm_colscan_kernel(uint32_t *cnt, uint32_t nb, uint32_t nbw, uint32_t J, uint32_t *counts) {
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  uint32_t k = g / nbw, b = g % nbw;
  uint32_t *col = cnt + (size_t)k * J * nbw + b;
  uint32_t run = 0;
  for (uint32_t j = 0; j < J; j++) {
    uint32_t v = col[(size_t)j * nbw];
    col[(size_t)j * nbw] = run;
    run += v;
  }
  counts[g] = run;
}

// entries[pos] = (point index << 1) | sign, grouped by bucket: LDS cursors = bucket offset + chunk prefix
static __global__ __launch_bounds__(SORT_THREADS) void msm_scatter_kernel(const uint16_t *dig, size_t stride, const uint32_t *sidx, uint32_t n, int K, uint32_t nbw,
                                                                        uint32_t J, uint32_t chunk, const uint32_t *cnt, const uint32_t *offsets, uint32_t *entries) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_bkt[];
  const uint32_t k = blockIdx.x % (uint32_t)K, j = blockIdx.x / (uint32_t)K;
  const uint32_t *pre = cnt + ((size_t)k * J + j) * nbw, *off = offsets + (size_t)k * nbw;
  for (uint32_t b = threadIdx.x; b < nbw; b += SORT_THREADS) s_bkt[b] = off[b] + pre[b];
  __syncthreads();
  const uint16_t *dk = dig + (size_t)k * stride;
  uint32_t i0 = j * chunk, i1 = min(i0 + chunk, n);
  for (uint32_t i = i0 + threadIdx.x; i < i1; i += SORT_THREADS) {
    uint32_t code = dk[sidx ? sidx[i] : i];
    if (code) {
      uint32_t pos = atomicAdd(&s_bkt[(code - 1u) >> 1], 1u);
      entries[pos] = (i << 1) | ((code - 1u) & 1u);
    }
  }
}

// exclusive scan of nb bucket counts in three launches (block sums -> scan of block sums -> apply);
// SCAN_BLOCK counts per workgroup.  Also lists buckets larger than big_thresh.
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER_THREAD = 8;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_PER_THREAD;

static __global__ __launch_bounds__(SCAN_THREADS) void msm_scan_sums_kernel(const uint32_t *counts, uint32_t nb, uint32_t *block_sums) {
  __shared__ uint32_t part[SCAN_THREADS];
  uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; k++) s += base + k < nb ? counts[base + k] : 0;
  part[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t off = SCAN_THREADS / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = part[0];
}

// in-place exclusive scan of nblocks block sums by one workgroup; writes the grand total to *total
static __global__ __launch_bounds__(1024) void msm_scan_top_kernel(uint32_t *block_sums, uint32_t nblocks, uint32_t *total) {
  __shared__ uint32_t part[1024];
  uint32_t t = threadIdx.x;
  uint32_t per = (nblocks + 1023) / 1024;
  uint32_t lo = t * per, hi = min(lo + per, nblocks);
  uint32_t s = 0;
  for (uint32_t b = lo; b < hi; b++) s += block_sums[b];
  part[t] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = part[t] - s;
  for (uint32_t b = lo; b < hi; b++) { uint32_t c = block_sums[b]; block_sums[b] = run; run += c; }
  if (t == 1023) *total = part[1023];
}

constexpr uint32_t BIG_MARK = 0xffffffffu;  // counts[b] after the scan: bucket b is owned by msm_big_kernel
static __global__ __launch_bounds__(SCAN_THREADS) void msm_scan_apply_kernel(uint32_t *counts, const uint32_t *block_sums, const uint32_t *total,
                                                                           uint32_t *offsets, uint32_t nb, uint32_t big_thresh,
                                                                           uint32_t *big_list, uint32_t *big_count, uint32_t big_cap) {
  __shared__ uint32_t part[SCAN_THREADS];
  uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
  uint32_t c[SCAN_PER_THREAD];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; k++) { c[k] = base + k < nb ? counts[base + k] : 0; s += c[k]; }
  part[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t off = 1; off < SCAN_THREADS; off <<= 1) {
    uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = block_sums[blockIdx.x] + part[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; k++) {
    if (base + k < nb) {
      offsets[base + k] = run;
      if (c[k] > big_thresh) {
        uint32_t slot = atomicAdd(big_count, 1u);
        if (slot < big_cap) { big_list[slot] = base + k; counts[base + k] = BIG_MARK; }  // beyond the cap the bucket stays with msm_accum_kernel
      }
      run += c[k];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) offsets[nb] = *total;
}

// bucket accumulation: thread per bucket.  Bucket sizes are ~Poisson(n / 2^(c-1)), so a wavefront would run
// at the pace of its fullest bucket; the workgroup therefore ranks its 256 buckets by size in LDS first
// and lane i takes the i-th largest, which makes the 64 lanes of a wavefront near-equal in trip count.
template <class F, int MINW>
static __global__ __launch_bounds__(MSM_THREADS, MINW) void msm_accum_kernel(const Affine<F> *points, const uint32_t *offsets, const uint32_t *entries,
                                                                     uint32_t nb, const uint32_t *counts, XYZZ<F> *buckets) {
  __shared__ uint32_t s_cnt[MSM_THREADS];
  __shared__ uint32_t s_order[MSM_THREADS];
  const uint32_t b0 = blockIdx.x * MSM_THREADS, tid = threadIdx.x;
  uint32_t my = 0;  // size of bucket b0 + tid; empty / out-of-range / msm_big_kernel-owned buckets sort last
  if (b0 + tid < nb) {
    uint32_t c = counts[b0 + tid];
    my = c == BIG_MARK ? 0 : c;
  }
  s_cnt[tid] = my;
  __syncthreads();
  uint32_t rank = 0;
  for (uint32_t j = 0; j < MSM_THREADS; j++) {
    uint32_t o = s_cnt[j];
    rank += (o > my) || (o == my && j < tid);
  }
  s_order[rank] = tid;
  __syncthreads();
  const uint32_t b = b0 + s_order[tid];
  if (b >= nb) return;
  const bool big = counts[b] == BIG_MARK;
  if (big) return;  // msm_big_kernel owns it
  uint32_t o0 = offsets[b], o1 = offsets[b + 1];
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint32_t j = o0; j < o1; j++) {
    uint32_t e = entries[j];
    Affine<F> p = load_pod(points + (e >> 1));
    acc = add_mixed(acc, p, (e & 1) != 0);
  }
  store_pod(buckets + b, acc);
}

// oversized buckets (0/1-heavy witnesses): BIG_SPLIT workgroups share one bucket, each does a strided
// accumulation and an LDS tree; msm_big_finish_kernel adds the BIG_SPLIT partial sums.
constexpr int BIG_SPLIT = 8;
constexpr int BIG_SLOTS = 64;  // bucket slots per launch round (grid = BIG_SLOTS * BIG_SPLIT)

template <class F, int MINW>
static __global__ __launch_bounds__(MSM_THREADS, MINW) void msm_big_kernel(const Affine<F> *points, const uint32_t *offsets, const uint32_t *entries,
                                                                   const uint32_t *big_list, const uint32_t *big_count, uint32_t big_cap,
                                                                   XYZZ<F> *partials) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(smem);
  uint32_t nbig = min(*big_count, big_cap);
  uint32_t sub = blockIdx.x % BIG_SPLIT;
  for (uint32_t w = blockIdx.x / BIG_SPLIT; w < nbig; w += BIG_SLOTS) {
    uint32_t b = big_list[w];
    uint32_t o0 = offsets[b], o1 = offsets[b + 1];
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t j = o0 + sub * MSM_THREADS + threadIdx.x; j < o1; j += MSM_THREADS * BIG_SPLIT) {
      uint32_t e = entries[j];
      Affine<F> p = load_pod(points + (e >> 1));
      acc = add_mixed(acc, p, (e & 1) != 0);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = MSM_THREADS / 2; s > 0; s >>= 1) {
      if (threadIdx.x < s) sh[threadIdx.x] = add_full(sh[threadIdx.x], sh[threadIdx.x + s]);
      __syncthreads();
    }
    if (threadIdx.x == 0) store_pod(partials + (size_t)w * BIG_SPLIT + sub, sh[0]);
    __syncthreads();
  }
}

template <class F, int MINW>
static __global__ __launch_bounds__(64, MINW) void msm_big_finish_kernel(const XYZZ<F> *partials, const uint32_t *big_list, const uint32_t *big_count,
                                                                         uint32_t big_cap, XYZZ<F> *buckets) {
  uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= min(*big_count, big_cap)) return;
  XYZZ<F> acc = load_pod(partials + (size_t)w * BIG_SPLIT);
  for (int k = 1; k < BIG_SPLIT; k++) acc = add_full(acc, load_pod(partials + (size_t)w * BIG_SPLIT + k));
  store_pod(buckets + big_list[w], acc);
}

// per group of 2^glog buckets of one window: sum_j (digit value) * B_j
template <class F, int MINW>
static __global__ __launch_bounds__(MSM_THREADS, MINW) void msm_reduce_kernel(const XYZZ<F> *buckets, MsmGeom g, XYZZ<F> *group_out) {
  uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t groups_per_window = g.nbw >> g.glog;
  if (gid >= groups_per_window * (uint32_t)g.K) return;
  uint32_t k = gid / groups_per_window, t = gid % groups_per_window;
  uint32_t gs = 1u << g.glog;
  const XYZZ<F> *B = buckets + (size_t)k * g.nbw + (size_t)t * gs;
  XYZZ<F> run = XYZZ<F>::inf(), T = XYZZ<F>::inf();
  for (int j = (int)gs - 1; j >= 0; j--) {
    run = add_full(run, load_pod(B + j));
    T = add_full(T, run);
  }
  // bucket j of this group holds digit value t*gs + j + 1:  T = sum (j+1) B_j, run = sum B_j
  if (t) T = add_full(T, mul_small(run, t * gs));
  store_pod(group_out + gid, T);
}

// workgroup per window: add the group results
template <class F, int MINW>
static __global__ __launch_bounds__(MSM_THREADS, MINW) void msm_final_kernel(const XYZZ<F> *group_out, uint32_t groups_per_window, XYZZ<F> *window_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(smem);
  const XYZZ<F> *G = group_out + (size_t)blockIdx.x * groups_per_window;
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint32_t j = threadIdx.x; j < groups_per_window; j += MSM_THREADS) acc = add_full(acc, load_pod(G + j));
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t s = MSM_THREADS / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] = add_full(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) store_pod(window_out + blockIdx.x, sh[0]);
}

// ---------------------------------------------------------------- device-side setup
// out[i] = scalar[i] * G through a fixed-base table T[j][d] = d * 2^(8j) * G (j < 32, d < 256; d = 0 unused),
// converted to the affine Montgomery wire form (infinity -> x = 0, y = one; binarify.ts:92-102).
template <class F, int MINW>
static __global__ __launch_bounds__(MSM_THREADS, MINW) void fixed_base_kernel(const Affine<F> *table, const Fr *scalars, size_t n, Affine<F> *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  DigitIter it;
  it.init(scalars[i].v);
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int j = 0; j < 32; j++) {
    uint32_t d = it.s[0] & 255u;
#pragma unroll
    for (int k = 0; k < 7; k++) it.s[k] = (it.s[k] >> 8) | (it.s[k + 1] << 24);
    it.s[7] >>= 8;
    if (d) acc = add_mixed(acc, load_pod(table + j * 256 + d));
  }
  store_pod(out + i, to_affine(acc));
}

// out[j] = in[idx[j]]  (table compaction / permutation at key build)
template <class T>
static __global__ void gather_kernel(const T *in, const uint32_t *idx, size_t n, T *out) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  store_pod(out + j, load_pod(in + idx[j]));
}

// VALU roofline microbenchmark: dependent chains of Fq Montgomery products, 4 independent chains/lane
static __global__ __launch_bounds__(MSM_THREADS) void fq_mul_bench_kernel(Fq *io, int iters) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = load_pod(io + i), b = a, c = a, d = a;
  b.v[0] ^= 1; c.v[1] ^= 2; d.v[2] ^= 3;
  for (int k = 0; k < iters; k++) {
    a = mul(a, b); b = mul(b, c); c = mul(c, d); d = mul(d, a);
  }
  store_pod(io + i, add(add(a, b), add(c, d)));
}

}  // namespace zkr
