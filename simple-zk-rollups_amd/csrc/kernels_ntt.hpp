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
//   The butterflies run on 9 x 29-bit limbs, lazily reduced (field29.hpp; bounds: see NTT_H_* below); an element is 8
//   words in HBM and 9 in LDS.  LDS tile is limb-major (SoA): lane l touches bank (l mod 32) for every ds_read_b32 ->
//   conflict free.
//   Twiddles w_{2s}^x come from one table per key (w_{2m}^k, k<m, serves all spans up to m; its x 2^256 form also gives the
//   coset factors) plus a 32 KB table w_2048^k shared by every contiguous pass (L1/L2 resident).
#pragma once
#include <type_traits>
#include "curve.hpp"
#include "field29.hpp"

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
constexpr int NTT_TILE_LOG = 11;   // 2048 elements = 72 KB of LDS per workgroup (2 workgroups / CU)
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

// a butterfly twiddle as the product takes it: nine 29-bit limbs, canonical, x 2^261 (36 bytes: no unpacking in the passes,
// which load three twiddles per four butterflies)
struct Tw29 { uint32_t v[9]; };
__device__ __forceinline__ L29<Fr29, 2> load_tw29(const Tw29 *p) {
  L29<Fr29, 2> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = p->v[i];
  return r;
}

enum NttPre { PRE_NONE = 0, PRE_COSET = 1, PRE_MUL = 2 };

struct NttPassArgs {
  const Fr *in0;   // input (may alias out)
  const Fr *in1;   // second operand for PRE_MUL
  Fr *out;
  const Fr *tw;    // w_{2^(tlog+1)}^k table, x 2^256 (the coset factors of PRE_COSET)
  const Tw29 *tw29;  // the same powers x 2^261 as limbs, 2^tlog + 1 entries (entry 2^tlog = -1): the butterflies' twiddles
  const Tw29 *twl29; // w_2048^k x 2^261, 2^TWL_LOG + 1 entries
  int tlog;        // log2 of tw entries
  int L;           // transform size 2^L
  int lo, hi;      // stages [lo,hi)
  int wlog;        // log2 columns per tile (0 when lo == 0)
  int pre;         // NttPre
  int canon;       // last pass of a transform: store canonical residues (between passes values below 3 r are stored)
  int prio;        // wave priority (s_setprio) of the pass: the preparation chain paces the pipeline once the accumulations are fast
  // a SECOND transform of the same shape in the same launch (gridDim.z = 2: calcH's transforms come in pairs -- the A and the B
  // side, then the two products): half the launches of the preparation chain and twice the workgroups per launch, which is
  // what a 2^17 transform lacks to fill the chip
  const Fr *in0_b, *in1_b;
  Fr *out_b;
  // first workgroup of the pass (0 = all of it).  A pass over stages [lo, hi) only mixes elements inside aligned blocks of 2^hi,
  // so when only a range of the OUTPUT is wanted -- a shard's part of h (zkr_prove.hip calc_h_device) -- the passes below
  // the top one run on the blocks that cover the range and on nothing else
  uint32_t blk_off;
  // first pass of a transform that is one aligned BLOCK of a larger one (a shard's block of a split calcH, zkr_prove.hip
  // calc_h_split): the coset factor of position gi of the block is that of coefficient (bitrev(gi) << pre_shift) | pre_add of the
  // whole transform of 2^(L + pre_shift) points.  0, 0: the transform is the whole one.
  uint32_t pre_shift, pre_add;
  // CROSS passes (template parameter): the top hi - lo <= 3 stages of a transform whose 2^(hi - lo) row blocks live in DIFFERENT
  // buffers -- the blocks of a vector split over the shards of a proof, each on its owner's device (peer access), or a local
  // buffer holding this shard's columns of every block.  Row r, column cg (position within the block): element
  // x_in0[r][cg - x_in_sub], stored to x_out[r][cg - x_out_sub].  Twiddles come from the true position (r << lo) + cg.
  const Fr *x_in0[8], *x_in1[8];
  Fr *x_out[8];
  uint32_t x_in_sub, x_out_sub;
};

// The butterflies run on 9 x 29-bit limbs (field29.hpp), lazily reduced: a product is 205 instructions instead of the ~300
// of the 32-bit form with its carry words, an addition or subtraction 9 independent operations and one carry sweep, and
// there is no conditional subtraction inside a pass (the 32-bit pass spent 539 instructions per butterfly, this one ~290).
// Data are in STANDARD form, twiddles x 2^261: mul(data, twiddle) is the plain product.  In HBM a value stays 8 x 32-bit
// words (32 B, below 3 r between passes, canonical after the last pass); in LDS it is 9 words, limb-major as before.
// Value bounds (half moduli, the H of L29<., H>):
//   DIF (u, v) -> (u + v, (u - v) w): the sum doubles the bound, the product resets it to 3.  In a pair of stages only one of
//     the four outputs is a double sum (4 H): that one goes through barrett() (< 2.5 r).  Invariant: every LDS value <= NTT_H_DIF.
//   DIT (u, v) -> (u + v w, u - v w): each stage adds at most 4 to the bound of either output, nothing needs reducing inside
//     a pass: a value read at stage j is below NTT_H_IN + 4 j <= NTT_H_DIT, whatever the type of the expression that wrote it
//     (the LDS words carry no type: a read states the bound, as unpack29 does, and this schedule is what makes it true);
//     barrett() once per element when the pass stores.
// An inverse twiddle is w^-x = -(w^(N - x)) (table entry N = -1 makes x = 0 regular): the butterfly takes w^(N - x) and
// swaps the roles of its sum and difference instead of negating.
constexpr int NTT_H_IN = 6;                                 // bound of a value loaded from HBM (below 3 r)
constexpr int NTT_H_DIF = 6;
constexpr int NTT_H_DIT = NTT_H_IN + 4 * NTT_TILE_LOG;      // 50
static_assert(NTT_H_DIT + 8 <= 128 && 2 * (NTT_H_DIT + 8) <= 338, "a DIT pass of NTT_TILE_LOG stages must fit the bounds of field29.hpp");
static_assert(NTT_STRIDED_LOG <= NTT_TILE_LOG, "a strided pass has no more stages than a contiguous one");
using NttL = Fr29;

// one pass; grid = 2^L / tile, block = NTT_THREADS, dynamic LDS = 36 * tile bytes
template <bool DIF, bool INV, int NTT_THREADS, bool CROSS = false>
static __global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(const NttPassArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  constexpr int HL = DIF ? NTT_H_DIF : NTT_H_DIT;  // what a value read from LDS is stated to be below
  using V = L29<NttL, HL>;
  // the three pointers a paired launch swaps, apart from the argument block (CROSS indexes its row tables per lane: they stay
  // where the kernel arguments are)
  const NttPassArgs &a = a_in;
  const Fr *const in0 = blockIdx.z ? a.in0_b : a.in0, *const in1 = blockIdx.z ? a.in1_b : a.in1;
  Fr *const out = blockIdx.z ? a.out_b : a.out;
  if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
  const int nb = a.hi - a.lo;
  const uint32_t rows = 1u << nb, W = 1u << a.wlog;
  const uint32_t tile = rows << a.wlog;
  const uint32_t lb = (1u << a.lo) >> a.wlog;  // tiles per q
  const uint32_t bx = blockIdx.x + a.blk_off;
  const uint32_t q = bx / lb, l0 = (bx % lb) << a.wlog;
  const size_t base = ((size_t)blockIdx.y << a.L) + ((size_t)q << a.hi) + l0;  // blockIdx.y: transform of a fused batch (vectors end to end)

  auto lds_get = [&](uint32_t e) {
    V x;
#pragma unroll
    for (int k = 0; k < 9; k++) x.v[k] = lds[k * tile + e];
    return x;
  };
  auto lds_put = [&](uint32_t e, const auto &x) {  // DIF: the type proves the invariant; DIT: the stage schedule does (see above)
    static_assert(!DIF || std::remove_reference_t<decltype(x)>::bound <= HL, "DIF invariant broken");
#pragma unroll
    for (int k = 0; k < 9; k++) lds[k * tile + e] = x.v[k];
  };

  for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
    uint32_t r = e >> a.wlog, c = e & (W - 1);
    Fr x;
    if constexpr (CROSS) {
      const uint32_t ci = l0 + c - a.x_in_sub;
      x = load_fr(a.x_in0[r] + ci);
      if (a.pre == PRE_MUL) x = mul(x, load_fr(a.x_in1[r] + ci));
    } else {
      size_t gi = base + ((size_t)r << a.lo) + c;
      x = load_fr(in0 + gi);
      if (a.pre == PRE_COSET) {  // first pass of a transform: canonical input, 32-bit product with the x 2^256 table (once per element)
        uint32_t br = __brev((uint32_t)gi << (32 - a.L));  // coefficient index of a bit-reversed position (low L bits of gi: position within its transform)
        br = (br << a.pre_shift) | a.pre_add;
        x = mul(x, load_fr(a.tw + ((size_t)br << (a.tlog - a.L - a.pre_shift))));
      } else if (a.pre == PRE_MUL) {
        x = mul(x, load_fr(in1 + gi));
      }
    }
    lds_put(e, unpack29<NttL, DIF ? NTT_H_DIF : NTT_H_IN>(x.v));
  }
  __syncthreads();

  // twiddle of span 2^slog at offset x: w_{2s}^x, or for an inverse transform w_{2s}^(s - x) = -(w_{2s}^-x)
  auto twiddle = [&](int slog, uint32_t x) {
    const Tw29 *tab = slog <= TWL_LOG ? a.twl29 : a.tw29;
    const int tlog = slog <= TWL_LOG ? TWL_LOG : a.tlog;
    uint32_t idx = x << (tlog - slog);
    if (INV) idx = (1u << tlog) - idx;
    return load_tw29(tab + idx);
  };
  // Stages run in PAIRS on four elements held in registers (one LDS round trip and one barrier per two stages; index
  // math modelled in tests/test_ntt_plan.py `double`); an odd stage count starts with one single stage.
  auto dif = [&](const auto &u, const auto &v, const L29<NttL, 2> &w, auto &sum, auto &prod) {  // (u + v, (u - v) w)
    sum = add(u, v);
    // the difference only ever becomes a factor of the product with the (normalised) twiddle: no carry sweep (field29.hpp U29)
    if constexpr (INV) prod = mul(w, sub_loose(v, u)); else prod = mul(w, sub_loose(u, v));
  };
  auto dit = [&](const auto &u, const auto &v, const L29<NttL, 2> &w, auto &hi, auto &lo_) {    // (u + v w, u - v w)
    auto t = mul(v, w);
    auto s = add(u, t).template to<std::remove_reference_t<decltype(hi)>::bound>();
    auto d = sub(u, t);
    if constexpr (INV) { hi = d; lo_ = s; } else { hi = s; lo_ = d; }
  };
  auto single_stage = [&](int j) {
    const uint32_t sl = 1u << j;
    for (uint32_t b = threadIdx.x; b < (tile >> 1); b += NTT_THREADS) {
      uint32_t c = b & (W - 1), kk = b >> a.wlog;
      uint32_t r0 = ((kk >> j) << (j + 1)) | (kk & (sl - 1));
      uint32_t e0 = (r0 << a.wlog) + c, e1 = e0 + (sl << a.wlog);
      auto w = twiddle(j + a.lo, ((kk & (sl - 1)) << a.lo) + l0 + c);
      V u = lds_get(e0), v = lds_get(e1);
      if constexpr (DIF) {
        L29<NttL, 2 * HL> s;
        L29<NttL, 3> p;
        dif(u, v, w, s, p);
        lds_put(e0, barrett(s));
        lds_put(e1, p);
      } else {
        L29<NttL, HL + 4> x0, x1;
        dit(u, v, w, x0, x1);
        lds_put(e0, x0);
        lds_put(e1, x1);
      }
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
      auto wa = twiddle(j + a.lo, im), wb0 = twiddle(j + 1 + a.lo, im), wb1 = twiddle(j + 1 + a.lo, im + (sl << a.lo));
      V x00 = lds_get(e00), x01 = lds_get(e01), x10 = lds_get(e10), x11 = lds_get(e11);
      if constexpr (DIF) {  // stage j + 1 first
        L29<NttL, 2 * HL> s0, s1;
        L29<NttL, 3> p0, p1;
        dif(x00, x10, wb0, s0, p0);
        dif(x01, x11, wb1, s1, p1);
        L29<NttL, 4 * HL> ss;
        L29<NttL, 3> sp, pp;
        L29<NttL, 6> ps;
        dif(s0, s1, wa, ss, sp);
        dif(p0, p1, wa, ps, pp);
        lds_put(e00, barrett(ss));  // the one double sum of the four outputs
        lds_put(e01, sp);
        lds_put(e10, ps);
        lds_put(e11, pp);
      } else {              // stage j first
        L29<NttL, HL + 4> y00, y01, y10, y11;
        dit(x00, x01, wa, y00, y01);
        dit(x10, x11, wa, y10, y11);
        L29<NttL, HL + 8> z00, z10, z01, z11;
        dit(y00, y10, wb0, z00, z10);
        dit(y01, y11, wb1, z01, z11);
        lds_put(e00, z00);
        lds_put(e01, z01);
        lds_put(e10, z10);
        lds_put(e11, z11);
      }
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
    V x = lds_get(e);
    Fr o;
    if constexpr (DIF) {
      static_assert(NTT_H_DIF <= NTT_H_IN, "what a DIF pass stores is what the next pass states on loading");
      if (a.canon) pack29(canonical_small(x), o.v); else pack29(x, o.v);
    } else {
      auto y = barrett(x);  // < 2.5 r <= NTT_H_IN
      if (a.canon) pack29(canonical_small(y), o.v); else pack29(y, o.v);
    }
    if constexpr (CROSS) store_fr(a.x_out[r] + (l0 + c - a.x_out_sub), o);
    else store_fr(out + base + ((size_t)r << a.lo) + c, o);
  }
}

// the butterflies' twiddle tables from the x 2^256 ones: out[k] = T[k] x 2^5 (= w^k x 2^261), k < n; out[n] = -1 x 2^261
static __global__ void twiddle261_kernel(const Fr *T, uint32_t n, Tw29 *out) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n) return;
  Fr w = k < n ? load_fr(T + k) : neg(Fr::one());
  auto l = canonical_small(mul(unpack29<Fr29, 10>(w.v), const29<Fr29>(Fr29::TO261)));
#pragma unroll
  for (int i = 0; i < 9; i++) out[k].v[i] = l.v[i];
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
// zero / zero_words: a few counters the NEXT kernels of the stream accumulate into (the range counts of the digit records,
// zkr_prove.hip msm_digits_enqueue), cleared here by workgroup 0 instead of by a memset launch of their own.
static __global__ void ingest_kernel(const Fr *in, Fr *out, size_t n, uint32_t *zero = nullptr, uint32_t zero_words = 0) {
  if (blockIdx.x == 0)
    for (uint32_t k = threadIdx.x; k < zero_words; k += blockDim.x) zero[k] = 0;
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
// blockIdx.z = 1: the B side of the QAP in the same launch (its CSR arrays and output vector)
struct SpmvSide { const uint32_t *row_ptr, *col; const Fr *coef; Fr *out; const uint32_t *wide; uint32_t n_wide; };
// row0 / row1: the rows wanted (a shard of a split calcH evaluates its block of the domain; otherwise 0 and m)
static __global__ void spmv_kernel(SpmvSide sa, SpmvSide sb, const Fr *w, uint32_t m, uint32_t n, uint32_t row0, uint32_t row1) {
  const SpmvSide &sd = blockIdx.z ? sb : sa;
  const uint32_t *row_ptr = sd.row_ptr, *col = sd.col;
  const Fr *coef = sd.coef;
  Fr *out = sd.out;
  uint32_t c = row0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= row1) return;
  w += (size_t)blockIdx.y * n;    // blockIdx.y: witness of a fused batch
  out += (size_t)blockIdx.y * m;
  uint32_t k0 = row_ptr[c], k1 = row_ptr[c + 1];
  if (k1 - k0 > SPMV_WIDE) return;
  Fr acc = Fr::zero();
  for (uint32_t k = k0; k < k1; k++) acc = add(acc, mul(load_fr(coef + k), load_fr(w + col[k])));
  store_fr(out + c, acc);
}
// wide[i] = index of the i-th row wider than SPMV_WIDE; one wavefront per row, terms strided over the lanes, LDS tree
static __global__ __launch_bounds__(64) void spmv_wide_kernel(SpmvSide sa, SpmvSide sb, const Fr *w, uint32_t m, uint32_t n, uint32_t row0, uint32_t row1) {
  __shared__ uint32_t sh[8 * 64];
  const SpmvSide &sd = blockIdx.z ? sb : sa;
  const uint32_t *row_ptr = sd.row_ptr, *col = sd.col, *wide = sd.wide;
  const Fr *coef = sd.coef;
  Fr *out = sd.out;
  if (blockIdx.x >= sd.n_wide) return;
  w += (size_t)blockIdx.y * n;
  out += (size_t)blockIdx.y * m;
  const uint32_t c = wide[blockIdx.x], lane = threadIdx.x;
  if (c < row0 || c >= row1) return;
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
// pos0 / pos1: the positions wanted (a shard: its range of h; otherwise 0 and 2^L)
static __global__ void combine_h_kernel(const Fr *S, const Fr *D, Fr *h, const Fr *tw, int tlog, int L, Fr c1, Fr c2, uint32_t *zero, uint32_t zero_words,
                                        uint32_t pos0, uint32_t pos1) {
  if (blockIdx.x == 0 && blockIdx.y == 0)  // see ingest_kernel: the counters of h's digit records
    for (uint32_t k = threadIdx.x; k < zero_words; k += blockDim.x) zero[k] = 0;
  uint32_t pos = pos0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= pos1) return;
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
