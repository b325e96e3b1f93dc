// kernels_msm.hpp -- multi-scalar multiplication over G1 / G2 (Pippenger bucket method over precomputed
// window tables) and the fixed-base kernels of the device-side setup.
//
// Path: the five multiexps of websnark's groth16GenProof (call site
// /root/reference/operator/src/snarks/common.ts:29).  Integer VALU work only (v_mad_u64_u32): no MFMA.
//
// Every key table holds, next to each base point P_i, its multiples 2^(ck) P_i for the K = ceil(255/c) windows
// (built once at key load, msm_precompute_kernel; 288 GB of HBM make a 13x table cheap).  The signed c-bit
// digit d of window k of scalar i then contributes sign(d) * (2^(ck) P_i) to bucket |d| of ONE bucket set
// shared by all windows, so c can be 20 (K = 13 additions per point instead of 16 at c = 16) while the
// reduction still runs over 2^19 buckets, and no Horner over windows is left.
//
// Pipeline per MSM (no host round trip until the result point comes back):
//   digits   : signed c-bit digits of every scalar, once per scalar vector, as records split by bucket range
//   hist     : workgroup per (bucket range, chunk): bucket occupancies counted with LDS atomics
//   scan     : per-bucket prefix over chunks, then exclusive prefix sum -> bucket offsets; buckets larger
//              than `big_thresh` are listed
//   order    : bucket ids sorted by occupancy, fullest first (counting sort over size classes)
//   scatter  : (table index, sign) entries grouped by bucket, cursors in LDS (counting sort)
//   accum    : one thread per bucket in that order, XYZZ += affine point (8M+2S), points gathered from the key table
//   big      : workgroups share an oversized bucket (0/1-heavy witnesses), LDS tree of XYZZ sums
//   reduce   : sum_b b*B_b in three launches (group running sums, bit-subset sums, finish) -> the MSM result
// Arithmetic: table points and buckets are PACKED in memory (8 x 32-bit words per coordinate, curve.hpp layouts) with
// coordinates x * 2^261 mod p; in registers the kernels work on 9 x 29-bit limbs with lazily reduced values (field29.hpp,
// curve29.hpp).  msm_precompute converts a freshly built table from the key's radix 2^256; msm_reduce3_kernel converts the
// result back for the host assembly.
#pragma once
#include "curve.hpp"
#include "curve29.hpp"  // the group law the accumulation / reduction kernels run: 9 x 29-bit limbs, radix 2^261 (field29.hpp)

namespace zkr {

#define ZKR_RED_PRIO 3  // wave priority of the oversized-bucket and reduction kernels (s_setprio takes a literal)
constexpr int MSM_THREADS = 256;
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

// 16-byte pieces straight from one place to another (global or LDS): a copy through a local T goes through the stack
template <class T>
__device__ __forceinline__ void copy_pod(T *dst, const T *src) {
  static_assert(sizeof(T) % 16 == 0, "16-byte granules");
  const uint4 *q = reinterpret_cast<const uint4 *>(src);
  uint4 *d = reinterpret_cast<uint4 *>(dst);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = q[i];
}

struct MsmGeom {
  uint32_t n;        // points (base points of the table; the table holds K * n)
  int c;             // window bits
  int K;             // windows
  uint32_t nbw;      // buckets = 2^(c-1), one set shared by all windows
  uint32_t big_thresh;
  int glog;          // reduce group = 2^glog buckets
  uint32_t S;        // reduce2: workgroups per task
  uint32_t batch;    // proofs fused into the launches (small circuits): `batch` bucket sets of nbw buckets laid end to end
};

// Signed-digit recoding state: scalar kept in 9 registers and shifted right by c each window so the
// digit is always the low bits (no runtime-indexed register array -> no scratch).
//
// Top-window spreading (round 3).  K = ceil(255 / c) windows cover c K bits, but a scalar has 254: at c = 20 the top window only
// ever sees 14 bits, so its 2^20 digits pile onto the 12 400 lowest buckets -- 105 entries in each of them against a mean of 26,
// and since a bucket is one thread's chain, those 194 wavefronts WERE the accumulation kernel: 825 us for a G1 table, whatever
// its instruction count or occupancy (profiles/r3_ab_top_window_spreading.txt).  A multiple of the group order changes no
// product: s P = (s + t r) P for a point of order r (every point of G1; the key's G2 points, which the setup makes multiples of
// the generator).  So a scalar whose top window is in play is recoded as s + t r with t = (its index) mod tmax, tmax = the
// largest count for which (t + 1) r still fits c K - 1 bits (42 at c = 20: no signed digit overflows): the top digits then spread
// over the whole bucket range (13.5 / 13 of the mean at most).  Scalars below 2^(c (K - 1)) -- zeros, booleans, 64-bit values --
// stay as they are: their top digit is zero, and as s + t r they would cost K additions instead of one.
struct DigitIter {
  uint32_t s[9];
  uint32_t carry;
  __device__ __forceinline__ void init(const uint32_t *p, uint32_t index = 0, uint32_t tmax = 0, int top_bit = 256) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    s[8] = 0;
    carry = 0;
    if (tmax > 1 && top_bit < 256) {
      // top window in play: any bit at or above top_bit = c (K - 1)
      const int tw = top_bit >> 5, to = top_bit & 31;
      uint32_t hi = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) hi |= i > tw ? s[i] : (i == tw ? s[i] >> to : 0u);
      const uint32_t t = hi ? index % tmax : 0u;
      uint64_t cy = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        cy += (uint64_t)s[i] + (uint64_t)t * FrParams::P[i];
        s[i] = (uint32_t)cy;
        cy >>= 32;
      }
      s[8] = (uint32_t)cy;
    }
  }
  __device__ __forceinline__ bool is_zero() const {
    return (s[0] | s[1] | s[2] | s[3] | s[4] | s[5] | s[6] | s[7] | s[8] | carry) == 0;
  }
  // returns signed digit in [-2^(c-1), 2^(c-1)]
  __device__ __forceinline__ int next(int c) {
    uint32_t raw = (s[0] & ((1u << c) - 1)) + carry;
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = (s[i] >> c) | (s[i + 1] << (32 - c));
    s[8] >>= c;
    int d = (int)raw;
    if (raw > (1u << (c - 1))) { d -= (1 << c); carry = 1; } else { carry = 0; }
    return d;
  }
};
// tmax of DigitIter::init for a window geometry: the largest t count with (t + 1) r <= 2^(c K - 1), capped (0 or 1: no spreading)
inline uint32_t digit_spread_tmax(int c, int K) {
  const int bits = c * K - 1;  // s + t r must stay below 2^bits so that the top signed digit does not overflow
  if (bits <= 254) return 0;
  if (bits - 254 >= 12) return 4096;
  // floor(2^bits / r) for bits in [255, 265]: r = 0x30644e72e131a029... * 2^192, i.e. 2^254 / r = 1.3226...
  const double ratio = 1.3225;  // 2^254 / r = 1.32253..., rounded DOWN
  uint32_t t = (uint32_t)(ratio * (double)(1u << (bits - 254)));
  return t;
}

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// ---------------------------------------------------------------- digit sort
// Bucket id = |d| - 1 < 2^(c-1) of the signed c-bit digit d; the 2^(c-1) buckets are cut into nR ranges of nbl
// <= 2^13 buckets, the number of 4-byte counters one workgroup keeps in LDS.
//
// Stage 1, once per SCALAR VECTOR (w serves the A, B1, B2 and C tables, h serves H): every non-zero digit
// becomes a record in the list of its bucket range -- ent_s = scalar index, ent_b = window << 17 | (bucket
// within the range) << 1 | (d < 0).  Two launches: count per list, then scatter (digits are recomputed; a
// scalar is 32 bytes and the recoding a few shifts).
//
// Lists and write combining (round 3).  A range's list is the concatenation of DIGIT_XCDS sub-lists, one per XCD slot
// (x = workgroup index mod 8, the round-robin placement of workgroups on the eight XCDs), and a workgroup reserves its span
// of sub-list (r, x) with one atomic: neighbouring spans then belong to workgroups of the SAME XCD, whose L2 completes the
// 64-byte lines they share before they leave.  The workgroup first groups its records by range in LDS (DIGIT_STAGE records:
// spt x 256 scalars x K digits) and then writes every run with consecutive lanes -- with 256 lists a workgroup writing in
// arrival order keeps 256 lines open for its whole life, a few thousand of them per XCD: more than its L2 holds, and the
// lines left in pieces (WRITE_SIZE 469 MB for 108 MB of records; profiles/r3_ab_sort_ranges.md).  Vectors whose K is too
// large for the stage (tiny circuits: K = 64 at c = 4) take the unstaged form of the same kernel.
// A workgroup takes spt x 256 scalars: every workgroup ends with one global atomic per range on the same counters, and those
// -- 262 k at 2^20 with one scalar per thread -- were the count kernel's bound (81 us against 30 at spt = 4).
constexpr int ENT_WIN_SHIFT = 17;
constexpr int ENT_TAG_SHIFT = 24;     // staged records carry their list's range in bits 24..31 until they are written out
static_assert(MSM_THREADS == 256, "the digit kernels keep one range counter per thread");
constexpr uint32_t MAX_RANGES = 256;  // x SORT_RANGE_MAX buckets = 2^21: window sizes up to c = 22 (2^24-point tables), or fused batches of small circuits
constexpr uint32_t DIGIT_XCDS = 8;
constexpr uint32_t DIGIT_STAGE = 2 * 256 * 13;  // records a workgroup groups in LDS (x 8 bytes = 53 KB)
constexpr uint32_t DIGIT_CLEAR_WORDS = 2 * DIGIT_XCDS * MAX_RANGES;  // counts[range][XCD slot] | fill cursors[range][XCD slot]: zero before the count kernel
constexpr size_t DIGIT_RNG_WORDS = 2 * DIGIT_XCDS * MAX_RANGES + MAX_RANGES + 1;  // counts[r][x] | fill cursors[r][x] | range offsets[nR + 1]

__device__ __forceinline__ uint32_t digit_bucket(int d) { return (uint32_t)(d < 0 ? -d : d) - 1u; }

// Fused batches (small circuits, DESIGN.md 3.2 "several proofs per launch"): `scalars` holds the vectors of several proofs
// end to end, n_per scalars each; proof p owns the bucket ranges [p * nR1, (p + 1) * nR1) -- i.e. its own bucket set --
// so one sort / accumulation / reduction launch serves every proof of the batch.  One proof: n_per = n, nR1 = nR.
static __global__ __launch_bounds__(MSM_THREADS) void msm_digits_count_kernel(const Fr *scalars, uint32_t n, uint32_t n_per, int c, int K, int nbl_log, uint32_t nR1, uint32_t nR, uint32_t *rng_cnt, uint32_t tmax, int spt) {
  __shared__ uint32_t s_cnt[MAX_RANGES];
  if (threadIdx.x < nR) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int q = 0; q < spt; q++) {
    uint32_t i = (blockIdx.x * spt + q) * blockDim.x + threadIdx.x;
    if (i < n) {
      const uint32_t r0 = (i / n_per) * nR1;
      DigitIter it;
      it.init(scalars[i].v, i, tmax, c * (K - 1));
      for (int k = 0; k < K; k++) {
        int d = it.next(c);
        if (d != 0) atomicAdd(&s_cnt[r0 + (digit_bucket(d) >> nbl_log)], 1u);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < nR && s_cnt[threadIdx.x]) atomicAdd(&rng_cnt[threadIdx.x * DIGIT_XCDS + blockIdx.x % DIGIT_XCDS], s_cnt[threadIdx.x]);
}

// rng_cnt complete, rng_fill zeroed; same grid and spt as the count.  Block 0 also publishes rng_off[0..nR] (exclusive prefix
// over the ranges) for the table sorts.  STAGED: dynamic LDS of DIGIT_STAGE x 8 bytes, needs spt * 256 * K <= DIGIT_STAGE.
template <bool STAGED>
static __global__ __launch_bounds__(MSM_THREADS) void msm_digits_scatter_kernel(const Fr *scalars, uint32_t n, uint32_t n_per, int c, int K, int nbl_log, uint32_t nR1, uint32_t nR,
                                                                              const uint32_t *rng_cnt, uint32_t *rng_fill, uint32_t *rng_off,
                                                                              uint32_t *ent_s, uint32_t *ent_b, uint32_t tmax, int spt) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_stage[];  // STAGED: [DIGIT_STAGE] scalar indices, [DIGIT_STAGE] tagged records
  __shared__ uint32_t s_cnt[MAX_RANGES], s_base[MAX_RANGES], s_loc[MAX_RANGES];
  const uint32_t t = threadIdx.x, x = blockIdx.x % DIGIT_XCDS;
  if (t < nR) s_cnt[t] = 0;
  __syncthreads();
  for (int q = 0; q < spt; q++) {
    uint32_t i = (blockIdx.x * spt + q) * blockDim.x + t;
    if (i < n) {
      const uint32_t r0 = (i / n_per) * nR1;
      DigitIter it;
      it.init(scalars[i].v, i, tmax, c * (K - 1));
      for (int k = 0; k < K; k++) {
        int d = it.next(c);
        if (d != 0) atomicAdd(&s_cnt[r0 + (digit_bucket(d) >> nbl_log)], 1u);
      }
    }
  }
  __syncthreads();
  // range starts = exclusive prefix of the range totals (one range per thread); this workgroup's sub-list starts behind the
  // sub-lists of the lower XCD slots.  The same scan carries the workgroup's own counts (high half) -> start of each run in the stage
  uint32_t own = 0, before = 0;
  if (t < nR) {
    const uint4 *cq = reinterpret_cast<const uint4 *>(rng_cnt + t * DIGIT_XCDS);
    const uint4 lo = cq[0], hi = cq[1];
    const uint32_t cx[DIGIT_XCDS] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (uint32_t j = 0; j < DIGIT_XCDS; j++) { own += cx[j]; before += j < x ? cx[j] : 0u; }
  }
  const uint32_t mine = t < nR ? s_cnt[t] : 0u;
  s_base[t] = own;
  s_loc[t] = mine;
  __syncthreads();
  for (uint32_t off = 1; off < MAX_RANGES; off <<= 1) {
    uint32_t v = t >= off ? s_base[t - off] : 0u, w = t >= off ? s_loc[t - off] : 0u;
    __syncthreads();
    s_base[t] += v;
    s_loc[t] += w;
    __syncthreads();
  }
  const uint32_t start = s_base[t] - own, lstart = s_loc[t] - mine, ltotal = s_loc[MAX_RANGES - 1];
  __syncthreads();
  if (t < nR) {
    if (blockIdx.x == 0) {
      rng_off[t] = start;
      if (t == nR - 1) rng_off[nR] = start + own;
    }
    s_base[t] = start + before + (mine ? atomicAdd(&rng_fill[t * DIGIT_XCDS + x], mine) : 0u);
    s_loc[t] = lstart;
    s_cnt[t] = 0;
  }
  __syncthreads();
  const uint32_t lo_mask = (1u << nbl_log) - 1u;
  for (int q = 0; q < spt; q++) {
    uint32_t i = (blockIdx.x * spt + q) * blockDim.x + t;
    if (i < n) {
      const uint32_t r0 = (i / n_per) * nR1;
      DigitIter it;
      it.init(scalars[i].v, i, tmax, c * (K - 1));
      for (int k = 0; k < K; k++) {
        int d = it.next(c);
        if (d != 0) {
          const uint32_t b = digit_bucket(d), r = r0 + (b >> nbl_log);
          const uint32_t rec = ((uint32_t)k << ENT_WIN_SHIFT) | ((b & lo_mask) << 1) | (d < 0 ? 1u : 0u);
          const uint32_t slot = atomicAdd(&s_cnt[r], 1u);
          if (STAGED) {
            s_stage[s_loc[r] + slot] = i;
            s_stage[DIGIT_STAGE + s_loc[r] + slot] = rec | (r << ENT_TAG_SHIFT);
          } else {
            ent_s[s_base[r] + slot] = i;
            ent_b[s_base[r] + slot] = rec;
          }
        }
      }
    }
  }
  if (STAGED) {
    __syncthreads();
    for (uint32_t e = t; e < ltotal; e += MSM_THREADS) {  // consecutive lanes, consecutive addresses inside a run
      const uint32_t rec = s_stage[DIGIT_STAGE + e], r = rec >> ENT_TAG_SHIFT;
      const uint32_t pos = s_base[r] + (e - s_loc[r]);
      ent_s[pos] = s_stage[e];
      ent_b[pos] = rec & ((1u << ENT_TAG_SHIFT) - 1u);
    }
  }
}

// Stages 2 and 4, per TABLE: one workgroup per (bucket range r, chunk j of that range's record list), bucket
// counters / cursors in LDS (32 KB), so the 13.6 M increments of a 2^20-point MSM
// are LDS atomics instead of L2/fabric atomics.  rank[s] = index of the table's point for scalar s, or
// RANK_NONE when the table has none (point at infinity in the key: dropped at key build); rank == null means the
// identity map (the table has a point for every scalar), which saves the gather.
constexpr int SORT_THREADS = 256;
constexpr uint32_t SORT_RANGE_MAX = 8192;  // bucket counters per workgroup (x 4 B of LDS): small footprint, so these
                                           // memory/LDS-bound workgroups find room on CUs busy with an accumulation
constexpr uint32_t SORT_RANGE_DEFAULT = 2048;   // buckets per range and records per (range, chunk) workgroup that msm_plan aims
constexpr uint32_t SORT_CHUNK_RECORDS = 3328;   // for (zkr_key.hip: the L2 footprint of the scatter decides)
constexpr uint32_t RANK_NONE = 0xffffffffu;
constexpr int SORT_UNROLL = 8;

__device__ __forceinline__ void sort_chunk(const uint32_t *rng_off, uint32_t r, uint32_t j, uint32_t J, uint32_t &e0, uint32_t &e1) {
  uint32_t lo = rng_off[r], hi = rng_off[r + 1];
  uint32_t chunk = (hi - lo + J - 1) / J;
  e0 = min(lo + j * chunk, hi);
  e1 = min(e0 + chunk, hi);
}

// Workgroup -> (bucket range r, chunk j).  The scatter's 4-byte writes of ONE range land in one window of the entry array
// (its buckets' slots, ~850 KB at 2^20) from all J chunks of that range: a 64-byte line gets its sixteen entries from
// sixteen workgroups.  Workgroups go to the eight XCDs round robin by index, and each XCD has its own L2: with the plain
// mapping (r = index / J) the chunks of a range sit on all eight, every L2 holds the line partially and writes it back
// with a byte mask (522 MB of traffic for 54 MB of entries, rocprofv3 WRITE_SIZE, round 2).  So: all chunks of a
// range on the XCD (r mod 8), whose L2 then merges the line before it leaves.
__device__ __forceinline__ void sort_block_to_chunk(uint32_t b, uint32_t nR, uint32_t J, uint32_t &r, uint32_t &j) {
  if (nR % 8 == 0) {
    const uint32_t xcd = b % 8, q = b / 8;  // q-th workgroup of this XCD
    r = xcd + 8 * (q / J);
    j = q % J;
  } else {
    r = b / J;
    j = b % J;
  }
}

// What the later kernels of one table's sort accumulate into, cleared by the FIRST kernel of that sort (msm_hist_kernel, workgroup
// 0) instead of by memset launches in front of it: a launch boundary lies between the clearing and every use, and two launches
// fewer per sort are two gaps fewer in a chain of short dependent kernels (DESIGN.md "preparation chain").
struct SortScratch {
  uint32_t *big_count;             // [0] oversized buckets listed, [1] total entries
  uint32_t *size_hist;             // [2][SIZE_BINS]: size-class histogram, hand-out counters of msm_order_kernel
};

// cnt[(r * J + j) * nbl + b] = occupancy of bucket r * nbl + b within chunk j
static __global__ __launch_bounds__(SORT_THREADS) void msm_hist_kernel(const uint32_t *ent_s, const uint32_t *ent_b, const uint32_t *rng_off, const uint32_t *rank,
                                                                     uint32_t n_per, uint32_t nbl, uint32_t J, uint32_t *cnt, SortScratch z) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_bkt[];
  if (blockIdx.x == 0 && z.big_count) {
    if (threadIdx.x < 2) z.big_count[threadIdx.x] = 0;
    for (uint32_t i = threadIdx.x; i < 2 * 1024u; i += SORT_THREADS) z.size_hist[i] = 0;  // 2 * SIZE_BINS (defined below)
  }
  const uint32_t j = blockIdx.x % J, r = blockIdx.x / J;
  for (uint32_t b = threadIdx.x; b < nbl; b += SORT_THREADS) s_bkt[b] = 0;
  __syncthreads();
  uint32_t e0, e1;
  sort_chunk(rng_off, r, j, J, e0, e1);
  // SORT_UNROLL records per trip: the record loads and the dependent rank gathers of a trip are all issued
  // before the first LDS atomic, otherwise every record pays the full gather latency in turn
  for (uint32_t e = e0 + threadIdx.x; e < e1; e += SORT_THREADS * SORT_UNROLL) {
    uint32_t b[SORT_UNROLL], idx[SORT_UNROLL];
#pragma unroll
    for (int u = 0; u < SORT_UNROLL; u++) {
      uint32_t eu = e + u * SORT_THREADS;
      bool in = eu < e1;
      b[u] = in ? ent_b[eu] : 0u;
      idx[u] = in ? (rank ? rank[ent_s[eu] % n_per] : 0u) : RANK_NONE;  // rank == null: every scalar has its point (identity map); % n_per: fused batches
    }
#pragma unroll
    for (int u = 0; u < SORT_UNROLL; u++)
      if (idx[u] != RANK_NONE) atomicAdd(&s_bkt[(b[u] >> 1) & (SORT_RANGE_MAX - 1u)], 1u);
  }
  __syncthreads();
  uint32_t *out = cnt + ((size_t)r * J + j) * nbl;
  for (uint32_t b = threadIdx.x; b < nbl; b += SORT_THREADS) out[b] = s_bkt[b];
}

// per bucket: exclusive prefix over the J chunks (in place) and the bucket total -> counts[]
static __global__ __launch_bounds__(MSM_THREADS) void msm_colscan_kernel(uint32_t *cnt, uint32_t nb, uint32_t nbl, uint32_t J, uint32_t *counts) {
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  uint32_t r = g / nbl, b = g % nbl;
  uint32_t *col = cnt + (size_t)r * J * nbl + b;
  uint32_t run = 0;
  for (uint32_t j = 0; j < J; j++) {
    uint32_t v = col[(size_t)j * nbl];
    col[(size_t)j * nbl] = run;
    run += v;
  }
  counts[g] = run;
}

// entries[pos] = ((k * n + rank) << 1) | sign  -- index into the table's window levels and the sign of the digit --
// grouped by bucket: LDS cursors = bucket offset + chunk prefix
static __global__ __launch_bounds__(SORT_THREADS) void msm_scatter_kernel(const uint32_t *ent_s, const uint32_t *ent_b, const uint32_t *rng_off, const uint32_t *rank,
                                                                        uint32_t n_per, uint32_t n, uint32_t nbl, uint32_t J, const uint32_t *cnt, const uint32_t *offsets,
                                                                        uint32_t *entries) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_bkt[];
  uint32_t j, r;
  sort_block_to_chunk(blockIdx.x, gridDim.x / J, J, r, j);
  const uint32_t *pre = cnt + ((size_t)r * J + j) * nbl, *off = offsets + (size_t)r * nbl;
  for (uint32_t b = threadIdx.x; b < nbl; b += SORT_THREADS) s_bkt[b] = off[b] + pre[b];
  __syncthreads();
  uint32_t e0, e1;
  sort_chunk(rng_off, r, j, J, e0, e1);
  for (uint32_t e = e0 + threadIdx.x; e < e1; e += SORT_THREADS * SORT_UNROLL) {
    uint32_t b[SORT_UNROLL], idx[SORT_UNROLL];
#pragma unroll
    for (int u = 0; u < SORT_UNROLL; u++) {
      uint32_t eu = e + u * SORT_THREADS;
      bool in = eu < e1;
      b[u] = in ? ent_b[eu] : 0u;
      idx[u] = in ? (rank ? rank[ent_s[eu] % n_per] : ent_s[eu] % n_per) : RANK_NONE;
    }
#pragma unroll
    for (int u = 0; u < SORT_UNROLL; u++) {
      if (idx[u] != RANK_NONE) {
        uint32_t pos = atomicAdd(&s_bkt[(b[u] >> 1) & (SORT_RANGE_MAX - 1u)], 1u);
        entries[pos] = (((b[u] >> ENT_WIN_SHIFT) * n + idx[u]) << 1) | (b[u] & 1u);
      }
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

constexpr uint32_t SIZE_BINS = 1024;  // bucket size classes for the fullest-first ordering (sizes >= SIZE_BINS-1 share the top class)
__device__ __forceinline__ uint32_t size_bin(uint32_t c) { return c < SIZE_BINS - 1 ? c : SIZE_BINS - 1; }
constexpr uint32_t BIG_MARK = 0xffffffffu;  // counts[b] after the scan: bucket b is owned by msm_big_kernel
static __global__ __launch_bounds__(SCAN_THREADS) void msm_scan_apply_kernel(uint32_t *counts, const uint32_t *block_sums, const uint32_t *total,
                                                                           uint32_t *offsets, uint32_t nb, uint32_t big_thresh,
                                                                           uint32_t *big_list, uint32_t *big_count, uint32_t big_cap, uint32_t *size_hist) {
  __shared__ uint32_t part[SCAN_THREADS];
  __shared__ uint32_t s_hist[SIZE_BINS];
  for (uint32_t b = threadIdx.x; b < SIZE_BINS; b += SCAN_THREADS) s_hist[b] = 0;
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
      bool big = false;
      if (c[k] > big_thresh) {
        uint32_t slot = atomicAdd(big_count, 1u);
        if (slot < big_cap) { big_list[slot] = base + k; counts[base + k] = BIG_MARK; big = true; }  // beyond the cap the bucket stays with msm_accum_kernel
      }
      atomicAdd(&s_hist[big ? 0u : size_bin(c[k])], 1u);
      run += c[k];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) offsets[nb] = *total;
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < SIZE_BINS; b += SCAN_THREADS)
    if (s_hist[b]) atomicAdd(&size_hist[b], s_hist[b]);
}

static_assert(SIZE_BINS == 1024, "msm_hist_kernel clears 2 * 1024 words of size_hist");

// order[] = all bucket ids sorted by occupancy, fullest first (counting sort over SIZE_BINS size classes;
// msm_big_kernel-owned and empty buckets go last).  Same 2048-bucket partition as the scan.  size_hist is
// complete when this runs; `taken` (zeroed) hands out ranges inside each size class.
static __global__ __launch_bounds__(SCAN_THREADS) void msm_order_kernel(const uint32_t *counts, uint32_t nb, const uint32_t *size_hist, uint32_t *taken, uint32_t *order) {
  __shared__ uint32_t s_start[SIZE_BINS];  // first: global histogram -> start of each class (descending sizes)
  __shared__ uint32_t s_hist[SIZE_BINS];   // local histogram, then base of this workgroup's range in each class
  __shared__ uint32_t part[SCAN_THREADS];
  constexpr uint32_t PER = SIZE_BINS / SCAN_THREADS;
  const uint32_t t = threadIdx.x;
  // descending exclusive prefix: class SIZE_BINS-1 starts at 0
  uint32_t v[PER], sum = 0;
#pragma unroll
  for (uint32_t q = 0; q < PER; q++) { v[q] = size_hist[SIZE_BINS - 1 - (t * PER + q)]; sum += v[q]; }
  part[t] = sum;
  for (uint32_t b = t; b < SIZE_BINS; b += SCAN_THREADS) s_hist[b] = 0;
  __syncthreads();
  for (uint32_t off = 1; off < SCAN_THREADS; off <<= 1) {
    uint32_t u = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += u;
    __syncthreads();
  }
  uint32_t run = part[t] - sum;
#pragma unroll
  for (uint32_t q = 0; q < PER; q++) { s_start[SIZE_BINS - 1 - (t * PER + q)] = run; run += v[q]; }
  // local ranks
  uint32_t base = blockIdx.x * SCAN_BLOCK + t * SCAN_PER_THREAD;
  uint32_t bin[SCAN_PER_THREAD], rank[SCAN_PER_THREAD];
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; k++) {
    if (base + k < nb) {
      uint32_t c = counts[base + k];
      bin[k] = c == BIG_MARK ? 0u : size_bin(c);
      rank[k] = atomicAdd(&s_hist[bin[k]], 1u);
    }
  }
  __syncthreads();
  for (uint32_t b = t; b < SIZE_BINS; b += SCAN_THREADS) {
    uint32_t h = s_hist[b];
    if (h) s_hist[b] = s_start[b] + atomicAdd(&taken[b], h);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; k++)
    if (base + k < nb) order[s_hist[bin[k]] + rank[k]] = base + k;
}

// bucket accumulation: thread per bucket, buckets taken in `order` (fullest first): the 64 lanes of a
// wavefront get buckets of (almost) equal occupancy, so no lane idles while another finishes, and the long
// buckets start first.
// The chain is software-pipelined: the entry index runs TWO additions ahead and -- AHEAD: G1, 64-byte points -- the point ONE,
// so a gather's address has been in a register for a whole addition when the gather is issued; for the 128-byte G2 points a
// second point in registers would spill, there only the index runs ahead.  Loads past the end of a chain re-read its last
// entry / point (no branch around a load, nothing out of bounds).
// Round 6 measured around this loop: B1 + A + C in ONE launch (the launch tails it removes are where the other
// streams' kernels found room: -3 %), two points ahead, the next point through the LDS DMA path, aligned non-temporal loads, the
// G2 launch on a stream of its own -- all within +-1 % of this form; with every gather redirected into 1 MB of its table
// (garbage sums) the pipelined rate is 8.6 % higher: that is all the memory side costs (profiles/r6_07_fake_gather_bound.txt).
constexpr int ACC_ONTO = 1, ACC_ZERO_BIG = 2;  // flags of the accumulation kernels' `onto` argument
constexpr int ACC_THREADS = 256;
template <class F, int MINW, bool AHEAD>
static __global__ __launch_bounds__(ACC_THREADS) __attribute__((amdgpu_waves_per_eu(MINW, MINW))) void msm_accum_kernel(const Affine<F> *points, const uint32_t *offsets, const uint32_t *entries,
                                                                     uint32_t nb, const uint32_t *counts, const uint32_t *order, XYZZ<F> *buckets, int onto) {
  using C = typename CoordOf<F>::C;
  const uint32_t t = blockIdx.x * ACC_THREADS + threadIdx.x;
  if (t >= nb) return;
  const uint32_t b = order[t];
  // msm_big_kernel owns an oversized bucket.  ACC_ZERO_BIG: its slot is cleared here, because the partial sums will be ADDED
  // to it later (the first table of a shared bucket set: see ACC_ONTO)
  if (counts[b] == BIG_MARK) {
    if (onto & ACC_ZERO_BIG) store_pod(buckets + b, XYZZ<F>::inf());
    return;
  }
  uint32_t o0 = offsets[b], o1 = offsets[b + 1];
  // ACC_ONTO: the bucket set already holds another table's sums over the same bucket geometry (C before H: only C + H is
  // ever needed, so one bucket set and ONE reduction chain serve both tables)
  XYZZ29<C> acc = (onto & ACC_ONTO) ? unpack_xyzz(load_pod(buckets + b)) : XYZZ29<C>::inf();
  if (o0 < o1) {
    const uint32_t last = o1 - 1;
    uint32_t e = entries[o0], e1 = entries[min(o0 + 1, last)];
    if (AHEAD) {
      Affine<F> p = load_pod(points + (e >> 1));
      for (uint32_t j = o0; j < o1; j++) {
        const uint32_t e2 = entries[min(j + 2, last)];
        const Affine<F> pn = load_pod(points + (e1 >> 1));
        if (!p.is_inf()) acc = add_mixed29<C>(acc, unpack_affine(p), (e & 1) != 0);  // infinity: placeholder of a shared-support table
        e = e1; e1 = e2; p = pn;
      }
    } else {
      for (uint32_t j = o0; j < o1; j++) {
        const uint32_t e2 = entries[min(j + 2, last)];
        const Affine<F> p = load_pod(points + (e >> 1));
        if (!p.is_inf()) acc = add_mixed29<C>(acc, unpack_affine(p), (e & 1) != 0);
        e = e1; e1 = e2;
      }
    }
  }
  store_pod(buckets + b, pack_xyzz<F>(acc));
}

// The same for small bucket sets (circuits of 2^17 constraints and below: fewer buckets than the chip has lanes):
// SPLIT neighbouring lanes share one bucket, each adds every SPLIT-th entry, the partial sums meet in a butterfly of
// lane exchanges.  log2(SPLIT) full additions extra per bucket buy SPLIT times the wavefronts and chains 1/SPLIT as long.
template <class T>
__device__ __forceinline__ T lane_xor_words(const T &v, int mask) {
  static_assert(sizeof(T) % 4 == 0, "made of 32-bit words");
  T r;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(&v);
  uint32_t *dst = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; i++) dst[i] = (uint32_t)__shfl_xor((int)src[i], mask);
  return r;
}
template <class F, int MINW, int SPLIT>
static __global__ __launch_bounds__(ACC_THREADS) __attribute__((amdgpu_waves_per_eu(MINW, MINW))) void msm_accum_split_kernel(const Affine<F> *points, const uint32_t *offsets, const uint32_t *entries,
                                                                     uint32_t nb, const uint32_t *counts, const uint32_t *order, XYZZ<F> *buckets, int onto) {
  const uint32_t t = blockIdx.x * ACC_THREADS + threadIdx.x;
  const uint32_t slot = t / SPLIT, sub = t % SPLIT;
  if (slot >= nb) return;  // uniform over the SPLIT lanes of a bucket
  const uint32_t b = order[slot];
  if (counts[b] == BIG_MARK) {  // see msm_accum_kernel
    if ((onto & ACC_ZERO_BIG) && sub == 0) store_pod(buckets + b, XYZZ<F>::inf());
    return;
  }
  using C = typename CoordOf<F>::C;
  const uint32_t o0 = offsets[b], o1 = offsets[b + 1];
  XYZZ29<C> acc = (onto & ACC_ONTO) && sub == 0 ? unpack_xyzz(load_pod(buckets + b)) : XYZZ29<C>::inf();  // see msm_accum_kernel
  uint32_t e = o0 + sub < o1 ? entries[o0 + sub] : 0u;
  for (uint32_t j = o0 + sub; j < o1; j += SPLIT) {
    uint32_t en = j + SPLIT < o1 ? entries[j + SPLIT] : 0u;
    Affine<F> p = load_pod(points + (e >> 1));
    if (!p.is_inf()) acc = add_mixed29<C>(acc, unpack_affine(p), (e & 1) != 0);
    e = en;
  }
  for (int m = 1; m < SPLIT; m <<= 1) acc = add_full29<C>(acc, lane_xor_words(acc, m));
  if (sub == 0) store_pod(buckets + b, pack_xyzz<F>(acc));
}

// oversized buckets (0/1-heavy witnesses): BIG_SPLIT workgroups share one bucket, each does a strided
// accumulation and an LDS tree; msm_big_finish_kernel adds the BIG_SPLIT partial sums.
// one step of an LDS tree over packed sums
template <class F>
__device__ __forceinline__ XYZZ<F> tree_add(const XYZZ<F> &a, const XYZZ<F> &b) {
  using C = typename CoordOf<F>::C;
  return pack_xyzz<F>(add_full29<C>(unpack_xyzz(a), unpack_xyzz(b), [&]() { return unpack_xyzz(b); }));
}
constexpr int BIG_SPLIT = 8;
constexpr int BIG_SLOTS = 64;  // bucket slots per launch round (grid = BIG_SLOTS * BIG_SPLIT)

// The oversized-bucket and reduction kernels below are few, long-running wavefronts that must find room on SIMDs whose register
// file the accumulations already hold (two wavefronts of 176 VGPRs for G1, of 248 for G2, out of 512).  Since round 6 they FIT:
// with the group law's products in written order (curve29.hpp ZKR_PIN_ORDER) the G1 forms take 125-160 VGPRs -- beside BOTH
// accumulation wavefronts of a SIMD, a third resident wavefront that displaces nothing -- and the G2 forms 242-331, beside ONE
// (the Fq2 running sums of reduce1 took 394 before: its workgroups could only start on CUs that had drained completely, i.e. in
// the tail of an accumulation launch; profiles/r6_03_steady_timeline_r5.md).  No kernel spills; tests/test_kernel_resources.py
// holds the register counts.
template <class F>
__device__ __forceinline__ void msm_big_body(const Affine<F> *points, const uint32_t *offsets, const uint32_t *entries,
                                             const uint32_t *big_list, const uint32_t *big_count, uint32_t big_cap, XYZZ<F> *partials) {
  __builtin_amdgcn_s_setprio(ZKR_RED_PRIO);  // few long-running wavefronts on the critical path: win VALU arbitration against the bulk accumulation
  using C = typename CoordOf<F>::C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(smem);
  uint32_t nbig = min(*big_count, big_cap);
  uint32_t sub = blockIdx.x % BIG_SPLIT;
  for (uint32_t w = blockIdx.x / BIG_SPLIT; w < nbig; w += BIG_SLOTS) {
    uint32_t b = big_list[w];
    uint32_t o0 = offsets[b], o1 = offsets[b + 1];
    XYZZ29<C> acc = XYZZ29<C>::inf();
    for (uint32_t j = o0 + sub * MSM_THREADS + threadIdx.x; j < o1; j += MSM_THREADS * BIG_SPLIT) {
      uint32_t e = entries[j];
      Affine<F> p = load_pod(points + (e >> 1));
      if (!p.is_inf()) acc = add_mixed29<C>(acc, unpack_affine(p), (e & 1) != 0);
    }
    sh[threadIdx.x] = pack_xyzz<F>(acc);
    __syncthreads();
    for (uint32_t s = MSM_THREADS / 2; s > 0; s >>= 1) {
      if (threadIdx.x < s) sh[threadIdx.x] = tree_add(sh[threadIdx.x], sh[threadIdx.x + s]);
      __syncthreads();
    }
    if (threadIdx.x == 0) copy_pod(partials + (size_t)w * BIG_SPLIT + sub, sh);
    __syncthreads();
  }
}

template <class F>
__device__ __forceinline__ void msm_big_finish_body(const XYZZ<F> *partials, const uint32_t *big_list, const uint32_t *big_count,
                                                    uint32_t big_cap, XYZZ<F> *buckets, int onto) {
  __builtin_amdgcn_s_setprio(ZKR_RED_PRIO);
  uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= min(*big_count, big_cap)) return;
  using C = typename CoordOf<F>::C;
  XYZZ29<C> acc = unpack_xyzz(load_pod(partials + (size_t)w * BIG_SPLIT));
  for (int k = 1; k < BIG_SPLIT; k++) {
    const XYZZ<F> *src = partials + (size_t)w * BIG_SPLIT + k;
    acc = add_full29<C>(acc, unpack_xyzz(load_pod(src)), [&]() { return unpack_xyzz(load_pod(src)); });
  }
  if (onto) {  // the bucket keeps the other table's sum (msm_accum_kernel skipped it)
    const XYZZ<F> *src = buckets + big_list[w];
    acc = add_full29<C>(acc, unpack_xyzz(load_pod(src)), [&]() { return unpack_xyzz(load_pod(src)); });
  }
  store_pod(buckets + big_list[w], pack_xyzz<F>(acc));
}

// ---------------------------------------------------------------- bucket reduction: result = sum_b (b+1) * B_b
// Three short launches per MSM; g = 2^glog buckets per group, ng = nbw / g groups:
//   reduce1 : thread per group, running sums over its g buckets -> R_t = sum_j B_{t,j}, T_t = sum_j (j+1) B_{t,j}
//             so that result = sum_t T_t + g * sum_t t * R_t        (2 additions per bucket, chains of 2g)
//   reduce2 : workgroup per (task, split): Q_j = sum of the R_t whose index t has bit j set (log2 ng tasks)
//             and the two halves of sum_t T_t -- plain sums: strided accumulation + LDS tree, S splits per task
//   reduce3 : one workgroup: adds the splits, turns Q_j into 2^(j+glog) Q_j by doublings (one lane per task,
//             in parallel), and adds the tasks in an LDS tree -> the MSM result
// group_out: R[ng] then T[ng];  task_out: [ntask][S] with ntask = log2(ng) + 2.
//
template <class F>
__device__ __forceinline__ void msm_reduce1_body(const XYZZ<F> *buckets, MsmGeom g, XYZZ<F> *group_out) {
  __builtin_amdgcn_s_setprio(ZKR_RED_PRIO);
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t ng = (g.nbw >> g.glog) * g.batch, gs = 1u << g.glog;  // groups of every bucket set of the batch, end to end
  if (t >= ng) return;
  using C = typename CoordOf<F>::C;
  const XYZZ<F> *B = buckets + (size_t)t * gs;
  XYZZ29<C> run = unpack_xyzz(load_pod(B + gs - 1)), T = run;
  for (int j = (int)gs - 2; j >= 0; j--) {
    run = add_full29<C>(run, unpack_xyzz(load_pod(B + j)), [&]() { return unpack_xyzz(load_pod(B + j)); });
    T = add_full29<C>(T, run, [&]() { return run; });
  }
  store_pod(group_out + t, pack_xyzz<F>(run));
  store_pod(group_out + (size_t)ng + t, pack_xyzz<F>(T));
}

template <class F>
__device__ __forceinline__ void msm_reduce2_body(const XYZZ<F> *group_out, MsmGeom g, XYZZ<F> *task_out) {
  __builtin_amdgcn_s_setprio(ZKR_RED_PRIO);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(smem);
  const uint32_t nglog = (uint32_t)(g.c - 1 - g.glog), ng = 1u << nglog;
  const uint32_t j = blockIdx.x / g.S, q = blockIdx.x % g.S;
  const uint32_t proof = blockIdx.y;  // bucket set of the batch (gridDim.y = g.batch)
  using C = typename CoordOf<F>::C;
  const XYZZ<F> *R = group_out + (size_t)proof * ng;
  XYZZ29<C> acc = XYZZ29<C>::inf();
  if (j < nglog) {  // the u-th index with bit j set, u < ng/2; this split takes u in [u0, u1)
    const uint32_t half = ng / 2, u0 = (uint32_t)((uint64_t)half * q / g.S), u1 = (uint32_t)((uint64_t)half * (q + 1) / g.S);
    for (uint32_t u = u0 + threadIdx.x; u < u1; u += MSM_THREADS) {
      uint32_t t = ((u >> j) << (j + 1)) | (1u << j) | (u & ((1u << j) - 1u));
      acc = add_full29<C>(acc, unpack_xyzz(load_pod(R + t)), [&]() { return unpack_xyzz(load_pod(R + t)); });
    }
  } else {
    const XYZZ<F> *T = group_out + (size_t)ng * g.batch + (size_t)proof * ng;
    const uint32_t h = j - nglog, lo = h * ng / 2, len = (h + 1) * ng / 2 - lo;
    const uint32_t t0 = lo + (uint32_t)((uint64_t)len * q / g.S), t1 = lo + (uint32_t)((uint64_t)len * (q + 1) / g.S);
    for (uint32_t t = t0 + threadIdx.x; t < t1; t += MSM_THREADS) acc = add_full29<C>(acc, unpack_xyzz(load_pod(T + t)), [&]() { return unpack_xyzz(load_pod(T + t)); });
  }
  sh[threadIdx.x] = pack_xyzz<F>(acc);
  __syncthreads();
  for (uint32_t s = MSM_THREADS / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] = tree_add(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) copy_pod(task_out + (size_t)proof * gridDim.x + blockIdx.x, sh);
}

// one workgroup of MSM_THREADS lanes per bucket set; needs ntask * S <= MSM_THREADS (msm_plan guarantees it)
template <class F>
__device__ __forceinline__ void msm_reduce3_body(const XYZZ<F> *task_out, MsmGeom g, XYZZ<F> *result) {
  __builtin_amdgcn_s_setprio(ZKR_RED_PRIO);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(smem);
  const uint32_t nglog = (uint32_t)(g.c - 1 - g.glog), ntask = nglog + 2, S = g.S;
  const uint32_t j = threadIdx.x / S, q = threadIdx.x % S;
  task_out += (size_t)blockIdx.x * ntask * S;  // this bucket set's partial sums
  result += blockIdx.x;
  sh[threadIdx.x] = j < ntask ? load_pod(task_out + threadIdx.x) : XYZZ<F>::inf();
  __syncthreads();
  uint32_t s = 1;
  while (s < S) s <<= 1;
  for (s >>= 1; s > 0; s >>= 1) {  // the S splits of one task (S need not be a power of two)
    if (j < ntask && q < s && q + s < S) sh[threadIdx.x] = tree_add(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  using C = typename CoordOf<F>::C;
  XYZZ<F> packed = XYZZ<F>::inf();
  if (threadIdx.x < ntask) {  // lane = task
    XYZZ29<C> acc = unpack_xyzz(sh[threadIdx.x * S]);
    if (threadIdx.x < nglog)
      for (uint32_t d = 0; d < threadIdx.x + (uint32_t)g.glog; d++) acc = dbl_xyzz29<C>(acc);
    packed = pack_xyzz<F>(acc);
  }
  __syncthreads();
  sh[threadIdx.x] = packed;
  __syncthreads();
  for (s = 16; s > 0; s >>= 1) {  // ntask <= 32
    if (threadIdx.x < s) sh[threadIdx.x] = tree_add(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {  // back to the key's radix (2^256), canonical: the host assembly takes over
    copy_pod(result, sh);
    xyzz_to_256_at(result);  // in place: no stack frame for the one cold call of the kernel
  }
}

template <class F>
static __global__ __launch_bounds__(MSM_THREADS) void msm_big_kernel(const Affine<F> *points, const uint32_t *offsets, const uint32_t *entries, const uint32_t *big_list,
                                                                    const uint32_t *big_count, uint32_t big_cap, XYZZ<F> *partials) {
  msm_big_body<F>(points, offsets, entries, big_list, big_count, big_cap, partials);
}
template <class F>
static __global__ __launch_bounds__(64) void msm_big_finish_kernel(const XYZZ<F> *partials, const uint32_t *big_list, const uint32_t *big_count, uint32_t big_cap,
                                                                   XYZZ<F> *buckets, int onto) {
  msm_big_finish_body<F>(partials, big_list, big_count, big_cap, buckets, onto);
}
template <class F>
static __global__ __launch_bounds__(MSM_THREADS) void msm_reduce1_kernel(const XYZZ<F> *buckets, MsmGeom g, XYZZ<F> *group_out) { msm_reduce1_body<F>(buckets, g, group_out); }
template <class F>
static __global__ __launch_bounds__(MSM_THREADS) void msm_reduce2_kernel(const XYZZ<F> *group_out, MsmGeom g, XYZZ<F> *task_out) { msm_reduce2_body<F>(group_out, g, task_out); }
template <class F>
static __global__ __launch_bounds__(MSM_THREADS) void msm_reduce3_kernel(const XYZZ<F> *task_out, MsmGeom g, XYZZ<F> *result) { msm_reduce3_body<F>(task_out, g, result); }

// ---------------------------------------------------------------- window tables (key load)
// tbl[k * n + i] = 2^(ck) * tbl[i] for k = 1..K-1, affine, x 2^261; level 0 comes in the key's radix (x 2^256, the wire
// form) and is rewritten x 2^261 as well.  Thread per base point: a Jacobian doubling chain on the 29-bit limbs
// (curve29.hpp dbl_jac29), whose points are stored unnormalised as they are reached; per PRE_LEVELS levels ONE inversion of
// the product of their denominators (kept in `ztmp`: PRE_LEVELS x n coordinates) makes them affine.  Against one inversion
// per stored multiple on the 32-bit form (the first version): 17 against 50 ms per G1 table and 29 against 63 ms for G2 at 2^20;
// zkr_key_load_websnark of the tx circuit's provingKeyBin 85 -> 52 ms, of a 2^20 key 668 -> 533 ms.
// The groups have odd order, so a multiple of a finite point is finite (a point off the curve may hit Y = 0: its
// multiples come out as garbage, as before; nothing else is affected).
constexpr int PRE_LEVELS = 12;
template <class F, int MINW>
static __global__ __launch_bounds__(MSM_THREADS, MINW) void msm_precompute_kernel(Affine<F> *tbl, uint32_t n, int c, int K, F *ztmp) {
  using C = typename CoordOf<F>::C;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> p = load_pod(tbl + i);
  if (p.is_inf()) {  // placeholder of a table laid out over a shared support: infinity at every level
    for (int k = 1; k < K; k++) store_pod(tbl + (size_t)k * n + i, p);
    return;
  }
  radix_to_261_at(&tbl[i].x);  // in place (pointer arguments: no stack frame for the two cold calls)
  radix_to_261_at(&tbl[i].y);
  p = load_pod(tbl + i);
  Jac29<C> q{C::template unpack<2>(p.x).template to<JX>(), C::template unpack<2>(p.y).template to<JY>(), C::one().template to<JZ>()};
  for (int k0 = 1; k0 < K; k0 += PRE_LEVELS) {
    const int nl = K - k0 < PRE_LEVELS ? K - k0 : PRE_LEVELS;
    auto prod = C::one().template to<4>();  // product of the chunk's denominators
    for (int l = 0; l < nl; l++) {
      for (int b = 0; b < c; b++) q = dbl_jac29<C>(q);
      store_pod(tbl + (size_t)(k0 + l) * n + i, Affine<F>{C::template pack<JX>(q.x), C::template pack<JY>(q.y)});
      store_pod(ztmp + (size_t)l * n + i, C::template pack<JZ>(q.z));
      prod = mul(prod, q.z).template to<4>();
    }
    auto inv = inv29(prod);  // 1 / (Z_0 ... Z_{nl-1})
    for (int l = nl - 1; l >= 0; l--) {
      auto pre = C::one().template to<4>();  // Z_0 ... Z_{l-1}
      for (int j = 0; j < l; j++) pre = mul(pre, C::template unpack<JZ>(load_pod(ztmp + (size_t)j * n + i))).template to<4>();
      auto iz = mul(inv, pre);               // 1 / Z_l
      inv = mul(inv, C::template unpack<JZ>(load_pod(ztmp + (size_t)l * n + i))).template to<4>();
      auto iz2 = sqr(iz);
      auto iz3 = mul(iz2, iz);
      Affine<F> a = load_pod(tbl + (size_t)(k0 + l) * n + i);
      a.x = C::template pack<2>(canonical_small(mul(C::template unpack<JX>(a.x), iz2)));
      a.y = C::template pack<2>(canonical_small(mul(C::template unpack<JY>(a.y), iz3)));
      store_pod(tbl + (size_t)(k0 + l) * n + i, a);
    }
  }
}

// coordinates of `count` affine points between the key's radix (x 2^256, what msm_precompute_kernel and the wire format use)
// and the hot path's (x 2^261); infinity (x = 0) stays infinity
template <class F>
static __global__ __launch_bounds__(MSM_THREADS) void radix_convert_kernel(Affine<F> *pts, size_t count, int to261) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  if (pts[i].is_inf()) return;
  if (to261) { radix_to_261_at(&pts[i].x); radix_to_261_at(&pts[i].y); } else { radix_to_256_at(&pts[i].x); radix_to_256_at(&pts[i].y); }
}

// to_affine (curve.hpp) for a kernel, inlined down to the inversion's loop: an out-of-line call takes its argument by reference,
// i.e. from memory, and keeps the caller's live registers in a stack frame across it -- all the "scratch" fixed_base_kernel had
// (160 / 464 B per lane, tools/kernel_resources.py).  Setup path: the code size does not matter.
__device__ __forceinline__ Fq inv_leaf(const Fq &x) { return inv_inline(x); }
__device__ __forceinline__ Fq2 inv_leaf(const Fq2 &x) {
  Fq n = inv_inline(add(sqr(x.a), sqr(x.b)));
  return Fq2{mul(x.a, n), neg(mul(x.b, n))};
}
template <class F>
__device__ __forceinline__ Affine<F> to_affine_leaf(const XYZZ<F> &p) {
  if (p.is_inf()) return Affine<F>{F::zero(), F::one()};
  F zi = inv_leaf(p.zzz);                  // 1/ZZZ
  F inv_zz = mul(sqr(p.zz), sqr(zi));      // ZZ^3 = ZZZ^2  =>  1/ZZ = ZZ^2/ZZZ^2
  return Affine<F>{mul(p.x, inv_zz), mul(p.y, zi)};
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
  store_pod(out + i, to_affine_leaf(acc));
}

// out[j] = in[idx[j]]  (table compaction / permutation at key build)
// One 16-byte piece per thread (4 per G1 point, 8 per G2 point): consecutive lanes write consecutive pieces, and no lane ever
// holds a whole point (round 4's thread-per-point form spilled the 128-byte G2 struct: 132 B of scratch per lane for a copy).
template <class T>
static __global__ void gather_kernel(const T *in, const uint32_t *idx, size_t n, T *out) {
  constexpr unsigned PIECES = sizeof(T) / 16;
  static_assert(sizeof(T) % 16 == 0, "points are copied in 16-byte pieces");
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t j = t / PIECES;
  const unsigned i = (unsigned)(t % PIECES);
  if (j >= n) return;
  const uint32_t src = idx[j];
  uint4 v = make_uint4(0, 0, 0, 0);  // no source point: the point at infinity (x = 0)
  if (src != 0xffffffffu) v = reinterpret_cast<const uint4 *>(in + src)[i];
  reinterpret_cast<uint4 *>(out + j)[i] = v;
}

// VALU roofline microbenchmark: dependent chains of the hot path's Fq Montgomery product (9 x 29-bit limbs, field29.hpp),
// 4 independent chains per lane; `legacy` runs the 8 x 32-bit product of field.hpp instead (the round-1 multiplier)
static __global__ __launch_bounds__(MSM_THREADS) void fq_mul_bench_kernel(Fq *io, int iters, int legacy) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = load_pod(io + i), b = a, c = a, d = a;
  b.v[0] ^= 1; c.v[1] ^= 2; d.v[2] ^= 3;
  if (legacy) {
    for (int k = 0; k < iters; k++) {
      a = mul(a, b); b = mul(b, c); c = mul(c, d); d = mul(d, a);
    }
    store_pod(io + i, add(add(a, b), add(c, d)));
    return;
  }
  auto la = unpack29<Fq29, 4>(a.v), lb = unpack29<Fq29, 4>(b.v), lc = unpack29<Fq29, 4>(c.v), ld = unpack29<Fq29, 4>(d.v);
  for (int k = 0; k < iters; k++) {
    la = mul(la, lb).to<4>(); lb = mul(lb, lc).to<4>(); lc = mul(lc, ld).to<4>(); ld = mul(ld, la).to<4>();
  }
  Fq r;
  pack29(weak(add(add(la, lb), add(lc, ld))), r.v);
  store_pod(io + i, r);
}

// Limb-level self test of the product forms of field29.hpp as the DEVICE runs them (one asm statement each,
// field29_asm.hpp): record = 8 operands x 9 raw limbs (a b c d e f g h), result = 9 raw limbs of
// form 0: a b   1: a^2   2: a b + c d   3: a b + c d + e f + g h    (x 2^-261 mod p, lazily reduced).
// tests/test_gpu_stages.py compares it bit for bit with the C++ recursion of the same header run on the host.
template <class PM>
static __global__ void f29_forms_kernel(const uint32_t *in, size_t n, int form, uint32_t *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t op[8][9], r[9];
#pragma unroll
  for (int k = 0; k < 8; k++)
#pragma unroll
    for (int j = 0; j < 9; j++) op[k][j] = in[i * 72 + k * 9 + j];
  f29_raw_form<PM>(form, op, r);
#pragma unroll
  for (int j = 0; j < 9; j++) out[i * 9 + j] = r[j];
}

}  // namespace zkr
