// zkr_key.hip -- proving-key ingestion: websnark binary -> device arena (CSR QAP rows, compacted
// Montgomery point tables, twiddle tables), replication hooks, workspace, fixed-base setup kernels.
//
// Replaces the key handling inside groth16GenProof (/root/reference/operator/src/snarks/common.ts:28-29);
// input layout is the one binarifyProvingKey writes (/root/reference/operator/src/utils/binarify.ts:143-206).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "kernels_msm.hpp"
#include "kernels_ntt.hpp"
#include "hostops.hpp"
#include "pairing.hpp"  // the curve equations (g1_on_curve / g2_on_curve constants) for ZKR_CHECK_POINTS
#include <map>
#include "zkr_internal.hpp"

namespace zkr {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// w_{2^k} in Montgomery form: 5^((r-1)/2^k) (5 = smallest quadratic non-residue mod r)
static Fr fr_root_of_unity(unsigned k) {
  Fr five = Fr::zero();
  five.v[0] = 5;
  Fr g = to_mont(five);
  uint32_t e[8];
  for (int i = 0; i < 8; i++) e[i] = FrParams::P[i];
  e[0] -= 1;
  // e >>= 28
  for (int i = 0; i < 8; i++) e[i] = (e[i] >> 28) | (i < 7 ? e[i + 1] << 4 : 0);
  Fr r = Fr::one(), b = g;
  for (int i = 0; i < 256; i++) {
    if ((e[i >> 5] >> (i & 31)) & 1) r = mul(r, b);
    b = sqr(b);
  }
  for (unsigned j = k; j < 28; j++) r = sqr(r);
  return r;
}
Fr host_root_of_unity(unsigned k) { return fr_root_of_unity(k); }

// Oversized buckets (summed by msm_big_kernel's workgroups instead of one thread of the accumulation).  A bucket of E entries is
// E dependent additions on one lane -- 7 us each for G1, 18 us for G2 -- so any bucket longer than the accumulation's bulk
// time stretches the whole kernel: the bulk is the launch's additions (n K per proof, nbat proofs in a fused launch) at ~100 k
// additions per chain step chip-wide, i.e. the threshold is n K nbat / 2^17, between twice and eight times the mean occupancy,
// never below 64.  Round 3's rule (max(8 mean, 256)) left buckets of up to 256 entries to single lanes of SPARSE bucket sets: a
// shard of a 2^22 key (mean 9) spent 4.9 ms in a G2 accumulation of 0.8 ms of work (profiles/r4_25_shard_big_threshold.txt).
uint32_t big_threshold(size_t n, int K, uint32_t nbw, int nbat) {
  const uint64_t mean = (uint64_t)n * K / nbw + 1;
  const uint64_t by_bulk = ((uint64_t)n * K * (uint64_t)(nbat < 1 ? 1 : nbat)) >> 17;
  uint64_t thr = by_bulk < mean * 8 ? by_bulk : mean * 8;
  if (thr < 2 * mean) thr = 2 * mean;  // dense little bucket sets (tiny circuits): only real outliers leave the accumulation
  return thr > 64 ? (uint32_t)thr : 64u;
}

// Window size from the length of the SCALAR vector (tables that share scalars share the digit codes,
// kernels_msm.hpp msm_digits_kernel) unless the key fixes it (c_fixed: the window tables in the arena were built
// for that c); chunking and the oversized-bucket threshold from the table itself.
MsmPlan msm_plan(size_t n_scalars, size_t n, int c_fixed) {
  MsmPlan pl;
  int lg = 0;
  while (((size_t)1 << lg) < n_scalars) lg++;
  int c = lg;  // 2^(c-1) buckets for ~n * 255/c entries: a few dozen entries per bucket
  if (c < 4) c = 4;
  // c stays at 20 above 2^20 scalars.  The digit sort takes up to 2^21 buckets (MAX_RANGES x SORT_RANGE_MAX), and by
  // multiplication counts c = 22 would pay at 2^24 points (12 instead of 13 additions per point for four times the
  // buckets: 1961 vs 2078 M multiplications per G1 table), but measured on the 2^24 rollup-shaped key it loses: 8.19 /
  // 7.50 proofs/s at c = 20 / 21, and at c = 22 the accumulation of a table takes 26.6 ms instead of 18 ms alone
  // (2 M bucket threads with short chains gather worse) and 3.5 -> 0.37 proofs/s with two proofs in flight; at 2^22
  // the counts already tie (530 / 544 / 532 M).  ZKR_MSM_C overrides (a documented knob: window bits of keys built in this process).
  if (c > 20) c = 20;
  if (const char *e = getenv("ZKR_MSM_C")) { int v = atoi(e); if (v >= 2 && v <= 22) c = v; }
  if (c_fixed) c = c_fixed;
  pl.c = c;
  pl.K = (255 + c - 1) / c;
  pl.nbw = 1u << (c - 1);
  pl.nb = pl.nbw;
  // bucket reduction in groups of 2^glog buckets (kernels_msm.hpp msm_reduce1_kernel: a chain of 2 * 2^glog - 2 additions
  // per thread).  At 2^19 buckets groups of 32 measured best (108 / 110 / 113 / 105 proofs/s at 8 / 16 / 32 / 64); with
  // fewer buckets the chip is not filled and the chain length is what counts: keep about 2^14 groups
  // (tx circuit, 2^16 buckets: 233 / 320 / 349 / 351 / 322 proofs/s at 32 / 16 / 8 / 4 / 2)
  int glog = c - 1 - 14 < 2 ? 2 : c - 1 - 14 > 5 ? 5 : c - 1 - 14;
  pl.glog = c - 1 < glog ? c - 1 : glog;
  pl.big_thresh = big_threshold(n, pl.K, pl.nbw, 1);
  // digit sort: one workgroup per (bucket range, chunk).  Ranges of 2048 buckets, chunks of ~3300 records: what counts is the
  // window of the entry array that the workgroups resident on one XCD scatter into together -- it has to stay in that XCD's
  // 4 MB L2 until its lines are complete (kernels_msm.hpp sort_block_to_chunk).  At 2^20 points: 256 ranges x 16 chunks, a
  // range's window is 212 KB, ~12 ranges in flight per XCD; with ranges of 8192 buckets (round 2) the same 16 chunks per range
  // kept 6.8 MB in flight per XCD and every line left L2 in pieces (WRITE_SIZE 386 MB per launch for 54 MB of entries, against
  // 98 MB now; kernel 159 -> 83 us; profiles/r3_ab_sort_ranges.md).
  uint32_t range_max = SORT_RANGE_DEFAULT;
  if (pl.nbw / range_max > MAX_RANGES) range_max = pl.nbw / MAX_RANGES;
  pl.nbl = pl.nbw < range_max ? pl.nbw : range_max;
  pl.nR = pl.nbw / pl.nbl;
  uint64_t J64 = ((uint64_t)n * pl.K + (uint64_t)pl.nR * SORT_CHUNK_RECORDS - 1) / ((uint64_t)pl.nR * SORT_CHUNK_RECORDS);
  uint32_t J = J64 > 64 ? 64u : (uint32_t)J64;
  if (J < 1) J = 1;
  pl.J = J;
  // reduction: every task sums ng/2 .. ng group results; one workgroup per 2048 of them, all tasks together at
  // most one workgroup per CU (msm_reduce3_kernel takes ntask * S <= MSM_THREADS partial sums)
  uint32_t ng = pl.nbw >> pl.glog, ntask = (uint32_t)(c - 1 - pl.glog) + 2;
  uint32_t S = (ng + 2047) / 2048;
  if (S > 16) S = 16;
  if (S > MSM_THREADS / ntask) S = MSM_THREADS / ntask;
  pl.S = S < 1 ? 1 : S;
  return pl;
}

// The arena's layout: header, twiddles, the two QAP sides (CSR), then per table its K window levels and its rank map.
// Everything that builds an arena -- key_build, the receiver of a compact arena, zkr_key_shard -- takes the offsets from here,
// so replicas and shards of one key agree byte for byte on where things are.
void arena_layout(ArenaHeader &h) {
  size_t off = ARENA_HEADER_BYTES;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return (uint64_t)o; };
  h.off_tw = take((size_t)h.m * 32);
  h.off_twl = take((size_t)(1u << TWL_LOG) * 32);
  for (int s = 0; s < 2; s++) {
    const uint32_t nnz = s == 0 ? h.nnzA : h.nnzB;
    h.off_rowptr[s] = take(((size_t)h.m + 1) * 4);
    h.off_col[s] = take((size_t)nnz * 4 + 4);
    h.off_coef[s] = take((size_t)nnz * 32 + 32);
    h.off_wide[s] = take((size_t)h.n_wide[s] * 4 + 4);
  }
  for (int t = 0; t < N_TABLES; t++) {
    const size_t pb = t == T_B2 ? 128 : 64, K = (255 + h.win_c[t] - 1) / h.win_c[t];
    h.off_pts[t] = take((size_t)h.npts[t] * K * pb + pb);  // K window levels per base point
    h.off_rank[t] = take((size_t)rank_entries(h, t) * 4 + 4);
  }
  h.total_len = off;
}

// A full arena's header against the layout its own sizes imply (arena_layout is the only producer): a damaged file or a foreign
// buffer must be refused here, not found by a kernel reading past a section.  Null when consistent, else what is wrong.
const char *arena_header_fault(const ArenaHeader &h, size_t len) {
  if (h.magic != ARENA_MAGIC) return "not a packed zkr key of this version (magic)";
  if (h.total_len != len) return "header total length differs from the bytes handed over";
  if (h.logm > 27 || h.m != (1u << h.logm) || h.n == 0 || h.p >= h.n) return "inconsistent circuit sizes";
  if (h.shard_parts < 1 || h.shard_parts > 64 || h.shard_part >= h.shard_parts) return "bad shard numbering";
  if (h.sc_n[0] > h.n || h.sc_lo[0] > h.n - h.sc_n[0] || h.sc_n[1] > h.m || h.sc_lo[1] > h.m - h.sc_n[1]) return "scalar ranges outside the vectors";
  if (h.shard_parts == 1 && (h.sc_lo[0] || h.sc_lo[1] || h.sc_n[0] != h.n || h.sc_n[1] != h.m)) return "a whole key whose scalar ranges are not the whole vectors";
  for (int s = 0; s < 2; s++)
    if (h.n_wide[s] > h.m) return "more wide rows than rows";
  for (int t = 0; t < N_TABLES; t++) {
    if (h.win_c[t] < 2 || h.win_c[t] > 26) return "bad window size";
    if (h.npts[t] > rank_entries(h, t)) return "more points than scalars in a table";
  }
  ArenaHeader want = h;
  arena_layout(want);
  bool same = want.total_len == h.total_len && want.off_tw == h.off_tw && want.off_twl == h.off_twl;
  for (int s = 0; s < 2; s++)
    same = same && want.off_rowptr[s] == h.off_rowptr[s] && want.off_col[s] == h.off_col[s] && want.off_coef[s] == h.off_coef[s] && want.off_wide[s] == h.off_wide[s];
  for (int t = 0; t < N_TABLES; t++) same = same && want.off_pts[t] == h.off_pts[t] && want.off_rank[t] == h.off_rank[t];
  return same ? nullptr : "section offsets differ from the layout the sizes imply";
}

// ... and the one size the layout rounds away: the CSR row pointers of both QAP sides end at the header's term counts
static int arena_rows_check(const unsigned char *arena, const ArenaHeader &h) {
  for (int s = 0; s < 2; s++) {
    uint32_t last = 0;
    ZKR_HIP_CHECK(hipMemcpy(&last, arena + h.off_rowptr[s] + (size_t)h.m * 4, 4, hipMemcpyDeviceToHost));
    const uint32_t nnz = s == 0 ? h.nnzA : h.nnzB;
    if (last != nnz) { set_error("arena: QAP side %d has %u terms by its row pointers, %u by the header", s, last, nnz); return ZKR_ERR_BAD_KEY; }
  }
  return 0;
}

// cap: proofs a fused batch can hold (cap bucket sets end to end; kernels_msm.hpp msm_digits_count_kernel)
// sets: bucket sets (with their reduction buffers) the workspace holds per proof -- 2 for B1, whose chain also reduces A's buckets
// (prove_submit_enqueue joint_ab)
// sets: bucket sets (and reduction buffers) per proof -- 2 in B1's workspace when A is accumulated behind B1's sets and reduced with
// them in one chain, 0 in A's own workspace then: it keeps only what its sort and its oversized buckets need
static int alloc_msm_ws(MsmWorkspace &ws, size_t n, const MsmPlan &pl, size_t xyzz_bytes, size_t cap = 1, size_t sets = 1) {
  size_t nb = pl.nb * cap;
  ZKR_HIP_CHECK(hipMalloc(&ws.counts, (nb + 1) * 4));
  ZKR_HIP_CHECK(hipMalloc(&ws.offsets, (nb + 1) * 4));
  ZKR_HIP_CHECK(hipMalloc(&ws.size_hist, 2 * SIZE_BINS * 4));
  ZKR_HIP_CHECK(hipMalloc(&ws.order, (nb + 1) * 4));
  ZKR_HIP_CHECK(hipMalloc(&ws.chunk_cnt, (size_t)pl.J * nb * 4 + 4));
  ZKR_HIP_CHECK(hipMalloc(&ws.entries, (n * pl.K * cap + 1) * 4));
  ZKR_HIP_CHECK(hipMalloc(&ws.big_list, BIG_CAP * 4));
  ZKR_HIP_CHECK(hipMalloc(&ws.big_count, 16));  // [0] oversized buckets, [1] total entries, [2] tile ticket (kernels_msm.hpp SortScratch)
  ZKR_HIP_CHECK(hipMalloc(&ws.block_sums, (nb / SCAN_BLOCK + 2) * 8));  // one 64-bit look-back word per scan tile (msm_scan_fused_kernel)
  ZKR_HIP_CHECK(hipMalloc(&ws.big_partials, (size_t)BIG_CAP * BIG_SPLIT * xyzz_bytes));
  if (sets) {
    ZKR_HIP_CHECK(hipMalloc(&ws.buckets, nb * sets * xyzz_bytes));
    // sized for the plan's groups AND for the smaller groups of a latency-mode chain (zkr_prove.hip msm_reduce_enqueue: groups of
    // 2^LAT_GLOG buckets when nothing else is in flight), whose task sums take up to 16 splits
    const int g_min = pl.glog < LAT_GLOG ? pl.glog : LAT_GLOG;
    ZKR_HIP_CHECK(hipMalloc(&ws.group_out, (size_t)(pl.nbw >> g_min) * cap * sets * 2 * xyzz_bytes));
    ZKR_HIP_CHECK(hipMalloc(&ws.task_out, (size_t)(pl.c + 2) * (pl.S > 16 ? pl.S : 16) * cap * sets * xyzz_bytes));
    ZKR_HIP_CHECK(hipMalloc(&ws.result, xyzz_bytes * cap * sets));
    ZKR_HIP_CHECK(hipHostMalloc(&ws.h_result, xyzz_bytes * cap * sets, hipHostMallocDefault));
  }
  ws.max_nb = nb;
  ws.max_entries = n * pl.K * cap;
  ws.sets = sets;
  return 0;
}
int digit_lists_alloc(DigitLists &dl, size_t n_scalars, const MsmPlan &pl) {  // n_scalars: of the whole (fused) vector
  ZKR_HIP_CHECK(hipMalloc(&dl.rng, DIGIT_RNG_WORDS * 4));
  ZKR_HIP_CHECK(hipMalloc(&dl.ent_s, (size_t)pl.K * n_scalars * 4 + 4));
  ZKR_HIP_CHECK(hipMalloc(&dl.ent_b, (size_t)pl.K * n_scalars * 4 + 4));
  return 0;
}
// How many proofs of this key one submit fuses into shared launches: circuits far below the size that fills the chip
// (the reference's own tx circuit: 2^17) run every kernel over `cap` witnesses at once -- vectors end to end, `cap` bucket
// sets per table -- so each launch has about the work of one 2^20 proof and the fixed tail of the bucket reduction is
// paid once per batch.  Bounds: the batch's bucket ranges must fit the digit sort's MAX_RANGES lists, the fused vectors
// stay at or below 2^20 elements, at most 16.
int fused_capacity(const ArenaHeader &h, const MsmPlan plan[N_TABLES]) {
  if (h.shard_parts > 1) return 1;  // a shard multiplies a sub-range of one proof's vectors: nothing to lay end to end
  uint32_t nr = 1;
  for (int t = 0; t < N_TABLES; t++) nr = plan[t].nR > nr ? plan[t].nR : nr;
  uint32_t cap = MAX_RANGES / nr;
  uint32_t by_size = h.m >= (1u << 20) ? 1u : (1u << 20) / h.m;
  if (cap > by_size) cap = by_size;
  if (cap > (uint32_t)MAX_FUSE) cap = MAX_FUSE;
  return cap < 1 ? 1 : (int)cap;
}
void digit_lists_free(DigitLists &dl) {
  hipFree(dl.rng); hipFree(dl.ent_s); hipFree(dl.ent_b);
  dl = DigitLists();
}
int msm_ws_alloc(MsmWorkspace &ws, size_t n, const MsmPlan &pl, size_t xyzz_bytes) { return alloc_msm_ws(ws, n, pl, xyzz_bytes, 1); }
void msm_ws_free(MsmWorkspace &ws) {
  hipFree(ws.counts); hipFree(ws.offsets); hipFree(ws.chunk_cnt); hipFree(ws.size_hist); hipFree(ws.order); digit_lists_free(ws.own_dig); hipFree(ws.entries); hipFree(ws.big_list); hipFree(ws.big_count); hipFree(ws.block_sums); hipFree(ws.big_partials);
  hipFree(ws.buckets); hipFree(ws.group_out); hipFree(ws.task_out); hipFree(ws.result);
  if (ws.h_result) hipHostFree(ws.h_result);
  ws = MsmWorkspace();
}

// The streams of a device, made ONCE per process and shared by every key on that device.  The HIP runtime multiplexes the streams
// of one priority over GPU_MAX_HW_QUEUES = 4 hardware queues and hands the queues out as streams come and go: a process that had
// created and freed a few keys (the bench's replication leg; an operator with its tx and withdraw keys and a key cache) found two
// streams of its NEXT key on one hardware queue -- preparation and a reduction chain serialised, 10 % off every rate (tx circuit
// 940 against 1 040 proofs/s, the real circuit at 2^20 133 against 154; with GPU_MAX_HW_QUEUES=8 they came back:
// tools/bench_leg_interference*.sh).  One set per device never moves.  Two keys proving at the same time share the streams: the
// launches of one proof are enqueued under the device's enqueue lock, so every cross-stream wait points at work enqueued before it
// and the shared streams cannot wait for each other in a circle.
namespace {
struct DeviceStreams {
  hipStream_t accum = nullptr, prep = nullptr, aux = nullptr, red[N_TABLES] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  std::mutex enqueue_mu;
};
std::mutex g_streams_mu;
std::map<int, DeviceStreams *> g_streams;  // never freed: the streams live as long as the process
}  // namespace

static int make_streams(DeviceStreams &ds, int n_red) {
  // The accumulation stream is the bulk; preparation and reduction chains are short dependent launches whose
  // delay stalls it (the next sort, the proof's completion), so their workgroups are dispatched first.
  int prio_lo = 0, prio_hi = 0;
  ZKR_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));  // numerically lower = higher priority
  if (!ds.accum) ZKR_HIP_CHECK(hipStreamCreateWithPriority(&ds.accum, hipStreamNonBlocking, prio_lo));
  if (!ds.prep) ZKR_HIP_CHECK(hipStreamCreateWithPriority(&ds.prep, hipStreamNonBlocking, prio_hi));
  for (int j = 0; j < n_red; j++)
    if (!ds.red[j]) ZKR_HIP_CHECK(hipStreamCreateWithPriority(&ds.red[j], hipStreamNonBlocking, prio_hi));
  // one more for C's oversized-bucket sums in a lone proof of a small circuit, and for a shard's witness-side sorts.  At the MIDDLE priority: a fifth
  // high-priority stream made the next one created -- the witness builder's (rollup_gpu.hip) -- share a hardware queue with the
  // preparation stream: its 28 ms kernel then stalled every proof behind it (facade pipeline 623 against 840 batches/s, tools/ab_hw_queues2.sh)
  if (!ds.aux) ZKR_HIP_CHECK(hipStreamCreateWithPriority(&ds.aux, hipStreamNonBlocking, (prio_lo + prio_hi) / 2));
  return 0;
}

int key_alloc_workspace(zkr_key *k) {
  ZKR_HIP_CHECK(hipSetDevice(k->device));
  if (int lrc = ntt_lds_check(k->device)) return lrc;
  const ArenaHeader &h = k->h;
  // reduction streams: the chains of one proof add up to ~6 ms of serialised launches, so on a single in-order
  // stream they, not the accumulations, set the pace with two proofs in flight (94 vs 106 proofs/s with two)
  for (int t = 0; t < N_TABLES; t++) k->plan[t] = msm_plan(rank_entries(h, t), h.npts[t], (int)h.win_c[t]);
  // Two reduction streams: the G2 chain on [0], the G1 chains one after the other on [1].  Round 2 ran circuits that fill the
  // chip alone with three (136.3 against 132.1 proofs/s at 2^20); since C and H share one chain and the G2 reductions take a
  // whole SIMD's registers, the third stream only adds contention for the accumulations: 152.5-153.5 against 145.9-146.6
  // proofs/s at 2^20 (three same-box rounds), 38.5 against 37.4 at 2^22, a synchronous 2^20 proof 7.75 against 8.0 ms; one
  // stream: 122 (tools/sweep_knobs2.sh).  Keys whose proofs are fused into shared launches always ran with two (fewer, fatter
  // chains: 2^16 1745 against 1680, 2^17 943 against 910, 2^18 514 against 492, 2^19 237 against 231, the tx circuit 988
  // against 957 proofs/s).
  k->n_red = 2;
  {
    DeviceStreams *ds = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_streams_mu);
      DeviceStreams *&slot = g_streams[k->device];
      if (!slot) slot = new DeviceStreams();
      ds = slot;
    }
    int src = 0;
    {
      std::lock_guard<std::mutex> lk(g_streams_mu);  // creation of missing streams of a shared set, one key at a time
      src = make_streams(*ds, k->n_red);
    }
    if (src) return src;
    k->streams_owner = ds;
    k->enqueue_mu = &ds->enqueue_mu;
    k->stream = ds->accum;
    k->prep_stream = ds->prep;
    for (int j = 0; j < k->n_red; j++) k->red_stream[j] = ds->red[j];
    k->red_stream[k->n_red] = ds->aux;  // an entry of red_stream behind the chains' streams
    k->n_all = k->n_red + 1;
    k->aux_stream = ds->aux;
  }
  {
    int rc = ntt_tables29_build((const Fr *)(k->arena + h.off_tw), 1u << h.tlog, (const Fr *)(k->arena + h.off_twl), 1u << TWL_LOG, nullptr, &k->tw29, &k->twl29);
    if (rc) return rc;
  }
  const size_t cap = (size_t)fused_capacity(h, k->plan);
  for (ProofSlot &sl : k->slot) {
    sl.cap = (int)cap;
    sl.rb.assign(32 * cap, 0);
    sl.sb.assign(32 * cap, 0);
    for (int t = 0; t < N_TABLES; t++) {
      ZKR_HIP_CHECK(hipEventCreateWithFlags(&sl.ev_done[t], hipEventDisableTiming));
      ZKR_HIP_CHECK(hipEventCreateWithFlags(&sl.ev_sorted[t], hipEventDisableTiming));
      ZKR_HIP_CHECK(hipEventCreateWithFlags(&sl.ev_res[t], hipEventDisableTiming));
    }
    ZKR_HIP_CHECK(hipEventCreateWithFlags(&sl.ev_w, hipEventDisableTiming));
    ZKR_HIP_CHECK(hipEventCreateWithFlags(&sl.ev_h, hipEventDisableTiming));
    for (int j = 0; j < k->n_all; j++) ZKR_HIP_CHECK(hipEventCreateWithFlags(&sl.ev_red[j], hipEventDisableTiming));
    ZKR_HIP_CHECK(hipMalloc(&sl.d_w, (size_t)h.n * 32 * cap));
    Fr **vecs[5] = {&sl.va, &sl.vb, &sl.ca, &sl.cb, &sl.d_h};
    for (auto v : vecs) ZKR_HIP_CHECK(hipMalloc(v, (size_t)h.m * 32 * cap));
    for (int t = 0; t < N_TABLES; t++) {
      const bool joint_ab = h.npts[T_A] && h.npts[T_B1] && same_reduce_geometry(k->plan[T_A], k->plan[T_B1]);  // A reduced with B1 (zkr_prove.hip)
      const size_t sets = joint_ab ? (t == T_B1 ? 2 : t == T_A ? 0 : 1) : 1;
      int rc = alloc_msm_ws(sl.ws[t], h.npts[t], k->plan[t], t == T_B2 ? sizeof(G2XYZZ) : sizeof(G1XYZZ), cap, sets);
      if (rc) return rc;
    }
    int rc = digit_lists_alloc(sl.dig_w, (size_t)h.sc_n[0] * cap, k->plan[T_A]);
    if (!rc) rc = digit_lists_alloc(sl.dig_h, (size_t)h.sc_n[1] * cap, k->plan[T_H]);
    if (rc) return rc;
  }
  // host-side window tables of delta_1 / delta_2 for the proof assembly (a few milliseconds; kept out of the first proof)
  std::call_once(k->delta_once, [&] {
    k->delta1_tab = fixed_base_table(load_g1(h.delta1));
    k->delta2_tab = fixed_base_table(load_g2(h.delta2));
  });
  return 0;
}

// `count` affine points between the key's Montgomery radix (2^256) and the hot path's (2^261), in place
int radix_convert(int device, bool g2, void *d_points, size_t count, bool to261) {
  if (count == 0) return 0;
  ZKR_HIP_CHECK(hipSetDevice(device));
  unsigned grid = (unsigned)((count + MSM_THREADS - 1) / MSM_THREADS);
  if (g2) radix_convert_kernel<Fq2><<<grid, MSM_THREADS>>>((G2Affine *)d_points, count, to261 ? 1 : 0);
  else radix_convert_kernel<Fq><<<grid, MSM_THREADS>>>((G1Affine *)d_points, count, to261 ? 1 : 0);
  ZKR_HIP_CHECK(hipGetLastError());
  ZKR_HIP_CHECK(hipDeviceSynchronize());
  return 0;
}

// Level 0 of the table (n points, the key's wire form: coordinates x 2^256) is in place: fills levels 1..K-1 and rewrites
// level 0, all in the radix of the accumulation kernels (x 2^261, canonical; kernels_msm.hpp header).
// ZKR_CHECK_POINTS=1: every base point of a table (wire form, x 2^256) satisfies its curve's equation y^2 = x^3 + b before
// anything is derived from it.  Off by default, as in the reference (websnark multiplies whatever the key holds): a point off
// the curve makes the window-table build produce garbage multiples silently (its doubling chain may hit Y = 0).
template <class F>
static __global__ void on_curve_kernel(const Affine<F> *pts, uint32_t n, F b, uint32_t *bad) {  // bad[0] = count, bad[1] = smallest index
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> p = load_pod(pts + i);
  if (p.is_inf()) return;  // placeholder of a shared-support table
  if (sqr(p.y) == add(mul(sqr(p.x), p.x), b)) return;
  atomicAdd(&bad[0], 1u);
  atomicMin(&bad[1], i);
}
static int check_points_on_curve(bool g2, const void *d_table, uint32_t n) {
  DevBuf buf;
  int rc = buf.alloc(8);
  if (rc) return rc;
  const uint32_t init[2] = {0u, 0xffffffffu};
  ZKR_HIP_CHECK(hipMemcpy(buf.p, init, 8, hipMemcpyHostToDevice));
  const unsigned grid = (n + 255) / 256;
  if (g2) on_curve_kernel<Fq2><<<grid, 256>>>((const G2Affine *)d_table, n, Fq2{pairing::fq_from_limbs(pairing::TWIST_B0), pairing::fq_from_limbs(pairing::TWIST_B1)}, buf.as<uint32_t>());
  else on_curve_kernel<Fq><<<grid, 256>>>((const G1Affine *)d_table, n, pairing::fq_small(3), buf.as<uint32_t>());
  ZKR_HIP_CHECK(hipGetLastError());
  uint32_t bad[2];
  ZKR_HIP_CHECK(hipMemcpy(bad, buf.p, 8, hipMemcpyDeviceToHost));
  if (bad[0]) {
    set_error("%u point(s) of a %s table are not on the curve (first: kept point %u); ZKR_CHECK_POINTS=1", bad[0], g2 ? "G2" : "G1", bad[1]);
    return ZKR_ERR_BAD_KEY;
  }
  return 0;
}

int msm_precompute(int device, bool g2, void *d_table, uint32_t n, const MsmPlan &pl) {
  if (n == 0) return 0;
  ZKR_HIP_CHECK(hipSetDevice(device));
  if (const char *e = getenv("ZKR_CHECK_POINTS"); e && atoi(e) != 0)
    if (int crc = check_points_on_curve(g2, d_table, n)) return crc;
  if (pl.K < 2) return radix_convert(device, g2, d_table, (size_t)n, true);
  const int levels = pl.K - 1 < PRE_LEVELS ? pl.K - 1 : PRE_LEVELS;
  DevBuf zbuf;  // denominators of one chunk of levels (msm_precompute_kernel)
  if (int arc = zbuf.alloc((size_t)n * levels * (g2 ? sizeof(Fq2) : sizeof(Fq)))) return arc;
  void *ztmp = zbuf.p;
  unsigned grid = (n + MSM_THREADS - 1) / MSM_THREADS;
  if (g2) msm_precompute_kernel<Fq2, 1><<<grid, MSM_THREADS>>>((G2Affine *)d_table, n, pl.c, pl.K, (Fq2 *)ztmp);
  else msm_precompute_kernel<Fq, 2><<<grid, MSM_THREADS>>>((G1Affine *)d_table, n, pl.c, pl.K, (Fq *)ztmp);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { set_error("window-table build failed: %s", hipGetErrorString(e)); return ZKR_ERR_HIP; }
  return 0;
}

// Builds the arena.  tbl_src[t]: full source table (host or device memory, affine Montgomery, 64/128 B
// per point); tbl_srcidx[t][j]: source index of kept point j; tbl_sidx[t][j]: index of its scalar.
int key_build(int device, uint32_t n, uint32_t p, uint32_t m, const std::vector<uint32_t> rowptr[2], const std::vector<uint32_t> col[2],
              const std::vector<uint8_t> coef[2], const void *const tbl_src[N_TABLES], const bool tbl_src_on_device[N_TABLES],
              const std::vector<uint32_t> tbl_srcidx[N_TABLES], const std::vector<uint32_t> tbl_sidx[N_TABLES], const uint8_t *consts448,
              zkr_key **out) {
  ZKR_HIP_CHECK(hipSetDevice(device));
  ArenaHeader h;
  memset(&h, 0, sizeof(h));
  h.magic = ARENA_MAGIC;
  h.n = n; h.p = p; h.m = m;
  h.logm = 0;
  while ((1u << h.logm) < m) h.logm++;
  h.tlog = h.logm;
  h.nnzA = (uint32_t)col[0].size();
  h.nnzB = (uint32_t)col[1].size();
  memcpy(h.alfa1, consts448, 64); memcpy(h.beta1, consts448 + 64, 64); memcpy(h.delta1, consts448 + 128, 64);
  memcpy(h.beta2, consts448 + 192, 128); memcpy(h.delta2, consts448 + 320, 128);
  h.sc_lo[0] = h.sc_lo[1] = 0; h.sc_n[0] = n; h.sc_n[1] = m;
  h.shard_part = 0; h.shard_parts = 1;
  std::vector<uint32_t> wide_rows[2];
  for (int s = 0; s < 2; s++) {
    for (uint32_t r = 0; r < m; r++)
      if (rowptr[s][r + 1] - rowptr[s][r] > SPMV_WIDE) wide_rows[s].push_back(r);
    h.n_wide[s] = (uint32_t)wide_rows[s].size();
  }
  // A and C multiply the same scalars and nearly the same signals (C lacks the public ones): when their supports
  // overlap by >= 90 % both tables are laid out over the UNION of the supports (a missing point is stored as
  // infinity, x = 0, and skipped by the accumulation), so one digit sort serves both (h.share_ac).
  const std::vector<uint32_t> *srcidx_eff[N_TABLES], *sidx_eff[N_TABLES];
  for (int t = 0; t < N_TABLES; t++) { srcidx_eff[t] = &tbl_srcidx[t]; sidx_eff[t] = &tbl_sidx[t]; }
  std::vector<uint32_t> u_sidx, uA_src, uC_src;
  {
    const std::vector<uint32_t> &sa = tbl_sidx[T_A], &sc = tbl_sidx[T_C];
    size_t i = 0, j = 0, both = 0;
    while (i < sa.size() || j < sc.size()) {
      uint32_t a = i < sa.size() ? sa[i] : 0xffffffffu, c = j < sc.size() ? sc[j] : 0xffffffffu;
      uint32_t s = a < c ? a : c;
      u_sidx.push_back(s);
      uA_src.push_back(a == s ? tbl_srcidx[T_A][i] : RANK_NONE);
      uC_src.push_back(c == s ? tbl_srcidx[T_C][j] : RANK_NONE);
      both += (a == s && c == s);
      if (a == s) i++;
      if (c == s) j++;
    }
    if (!u_sidx.empty() && both * 10 >= u_sidx.size() * 9) {
      h.share_ac = 1;
      srcidx_eff[T_A] = &uA_src; srcidx_eff[T_C] = &uC_src;
      sidx_eff[T_A] = sidx_eff[T_C] = &u_sidx;
    }
  }
  MsmPlan plan[N_TABLES];
  for (int t = 0; t < N_TABLES; t++) {
    h.npts[t] = (uint32_t)srcidx_eff[t]->size();
    plan[t] = msm_plan(t == T_H ? m : n, h.npts[t]);
    h.win_c[t] = (uint32_t)plan[t].c;
  }
  h.share_b = tbl_sidx[T_B1] == tbl_sidx[T_B2] ? 1 : 0;
  arena_layout(h);
  const size_t off = h.total_len;

  unsigned char *arena = nullptr;
  ZKR_HIP_CHECK(hipMalloc(&arena, off));
  ZKR_HIP_CHECK(hipMemset(arena, 0, off));  // alignment gaps and slack bytes are part of the arena's bytes (replicas, packed key files): defined, not stale
  ZKR_HIP_CHECK(hipMemcpy(arena, &h, sizeof(h), hipMemcpyHostToDevice));
  for (int s = 0; s < 2; s++) {
    ZKR_HIP_CHECK(hipMemcpy(arena + h.off_rowptr[s], rowptr[s].data(), rowptr[s].size() * 4, hipMemcpyHostToDevice));
    if (!col[s].empty()) {
      ZKR_HIP_CHECK(hipMemcpy(arena + h.off_col[s], col[s].data(), col[s].size() * 4, hipMemcpyHostToDevice));
      ZKR_HIP_CHECK(hipMemcpy(arena + h.off_coef[s], coef[s].data(), coef[s].size(), hipMemcpyHostToDevice));
    }
    if (!wide_rows[s].empty()) ZKR_HIP_CHECK(hipMemcpy(arena + h.off_wide[s], wide_rows[s].data(), wide_rows[s].size() * 4, hipMemcpyHostToDevice));
  }
  for (int t = 0; t < N_TABLES; t++) {
    size_t np = h.npts[t], pb = t == T_B2 ? 128 : 64;
    if (!np) continue;
    {  // scalar index -> point index (RANK_NONE: the key has the point at infinity there)
      std::vector<uint32_t> rank(t == T_H ? m : n, RANK_NONE);
      for (size_t j = 0; j < np; j++) rank[(*sidx_eff[t])[j]] = (uint32_t)j;
      bool ident = np == rank.size();
      for (size_t j = 0; ident && j < np; j++) ident = rank[j] == (uint32_t)j;
      h.rank_identity[t] = ident ? 1 : 0;
      ZKR_HIP_CHECK(hipMemcpy(arena + h.off_rank[t], rank.data(), rank.size() * 4, hipMemcpyHostToDevice));
    }
    if (tbl_src_on_device[t]) {
      uint32_t *d_idx = nullptr;
      ZKR_HIP_CHECK(hipMalloc(&d_idx, np * 4));
      ZKR_HIP_CHECK(hipMemcpy(d_idx, srcidx_eff[t]->data(), np * 4, hipMemcpyHostToDevice));
      unsigned grid = (unsigned)((np * (pb / 16) + 255) / 256);  // one 16-byte piece per thread
      if (t == T_B2)
        gather_kernel<G2Affine><<<grid, 256>>>((const G2Affine *)tbl_src[t], d_idx, np, (G2Affine *)(arena + h.off_pts[t]));
      else
        gather_kernel<G1Affine><<<grid, 256>>>((const G1Affine *)tbl_src[t], d_idx, np, (G1Affine *)(arena + h.off_pts[t]));
      ZKR_HIP_CHECK(hipGetLastError());
      ZKR_HIP_CHECK(hipDeviceSynchronize());
      hipFree(d_idx);
    } else {
      std::vector<uint8_t> stage(np * pb);
      const uint8_t *src = (const uint8_t *)tbl_src[t];
      for (size_t j = 0; j < np; j++) {
        uint32_t si = (*srcidx_eff[t])[j];
        if (si != RANK_NONE) memcpy(&stage[j * pb], src + (size_t)si * pb, pb);  // else: stays all-zero = infinity (x = 0)
      }
      ZKR_HIP_CHECK(hipMemcpy(arena + h.off_pts[t], stage.data(), np * pb, hipMemcpyHostToDevice));
    }
    int rc = msm_precompute(device, t == T_B2, arena + h.off_pts[t], (uint32_t)np, plan[t]);
    if (rc) { hipFree(arena); return rc; }
  }
  ZKR_HIP_CHECK(hipMemcpy(arena, &h, sizeof(h), hipMemcpyHostToDevice));  // again: rank_identity was filled in above
  // twiddles on device: T[k] = w_{2m}^k and w_2048^k
  twiddle_table_kernel<<<(m + 255) / 256, 256>>>((Fr *)(arena + h.off_tw), m, fr_root_of_unity(h.logm + 1));
  twiddle_table_kernel<<<((1u << TWL_LOG) + 255) / 256, 256>>>((Fr *)(arena + h.off_twl), 1u << TWL_LOG, fr_root_of_unity(TWL_LOG + 1));
  ZKR_HIP_CHECK(hipGetLastError());
  ZKR_HIP_CHECK(hipDeviceSynchronize());

  zkr_key *k = new zkr_key();
  k->device = device;
  k->arena = arena;
  k->arena_len = off;
  k->owns_arena = true;
  k->h = h;
  int rc = key_alloc_workspace(k);
  if (rc) { zkr_key_free(k); return rc; }
  *out = k;
  return 0;
}

// scalars (host, std form) * generator -> device array of affine Montgomery points (caller hipFree's).  wipe: the
// scalars are secret (setup with fresh toxic waste): their device copy is zeroed before it is freed, on every path.
template <class F>
static int fixed_base_impl(int device, const Affine<F> &gen, const uint8_t *scalars_std, size_t n, void **d_out, bool wipe) {
  ZKR_HIP_CHECK(hipSetDevice(device));
  // T[j][d] = d * 2^(8j) * G on the host (8160 group operations, once per generator)
  std::vector<Affine<F>> table(32 * 256);
  XYZZ<F> base = to_xyzz(gen);
  for (int j = 0; j < 32; j++) {
    XYZZ<F> cur = XYZZ<F>::inf();
    for (int d = 1; d < 256; d++) {
      cur = add_full(cur, base);
      table[j * 256 + d] = to_affine(cur);
    }
    table[j * 256] = Affine<F>{F::zero(), F::one()};
    for (int b = 0; b < 8; b++) base = dbl_xyzz(base);
  }
  struct DevBufs {  // freed (and the scalars cleared) however the function is left
    Affine<F> *table = nullptr, *pts = nullptr;
    Fr *sc = nullptr;
    size_t sc_bytes = 0;
    bool wipe = false;
    ~DevBufs() {
      if (sc) {
        if (wipe) { hipMemset(sc, 0, sc_bytes); hipDeviceSynchronize(); }
        hipFree(sc);
      }
      if (table) hipFree(table);
      if (pts) hipFree(pts);
    }
  } d;
  d.wipe = wipe;
  d.sc_bytes = (n + 1) * 32;
  ZKR_HIP_CHECK(hipMalloc(&d.table, table.size() * sizeof(Affine<F>)));
  ZKR_HIP_CHECK(hipMalloc(&d.pts, (n + 1) * sizeof(Affine<F>)));
  ZKR_HIP_CHECK(hipMalloc(&d.sc, d.sc_bytes));
  ZKR_HIP_CHECK(hipMemcpy(d.table, table.data(), table.size() * sizeof(Affine<F>), hipMemcpyHostToDevice));
  ZKR_HIP_CHECK(hipMemcpy(d.sc, scalars_std, n * 32, hipMemcpyHostToDevice));
  constexpr int MINW = sizeof(F) == 32 ? 2 : 1;
  fixed_base_kernel<F, MINW><<<(unsigned)((n + 255) / 256), 256>>>(d.table, d.sc, n, d.pts);
  ZKR_HIP_CHECK(hipGetLastError());
  ZKR_HIP_CHECK(hipDeviceSynchronize());
  *d_out = d.pts;
  d.pts = nullptr;  // handed to the caller
  return 0;
}

int fixed_base_points(int device, bool g2, const uint8_t *scalars_std, size_t n, void **d_out, bool wipe_scalars) {
  if (!g2) {
    G1Affine g{Fq::one(), add(Fq::one(), Fq::one())};  // (1, 2), TxVerifier.sol:24-26
    return fixed_base_impl<Fq>(device, g, scalars_std, n, d_out, wipe_scalars);
  }
  // G2 generator, snarkjs order [re, im] (TxVerifier.sol:30-35 lists [im, re])
  static const uint32_t GX0[8] = {0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu};
  static const uint32_t GX1[8] = {0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u};
  static const uint32_t GY0[8] = {0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u};
  static const uint32_t GY1[8] = {0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u};
  G2Affine g;
  memcpy(g.x.a.v, GX0, 32); memcpy(g.x.b.v, GX1, 32); memcpy(g.y.a.v, GY0, 32); memcpy(g.y.b.v, GY1, 32);
  g.x.a = to_mont(g.x.a); g.x.b = to_mont(g.x.b); g.y.a = to_mont(g.y.a); g.y.b = to_mont(g.y.b);
  return fixed_base_impl<Fq2>(device, g, scalars_std, n, d_out, wipe_scalars);
}

}  // namespace zkr

using namespace zkr;

extern "C" {

const char *zkr_last_error(void) { return g_err; }
const char *zkr_version(void) { return "zkr-hip 0.1 (gfx950)"; }

int zkr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int zkr_device_pci_bus_id(int device, char *out, size_t out_len) {
  if (!out || out_len < 16) { set_error("buffer too small"); return ZKR_ERR_ARG; }
  if (zkr_device_count() <= device || device < 0) { set_error("no HIP device %d", device); return ZKR_ERR_NO_DEVICE; }
  ZKR_HIP_CHECK(hipDeviceGetPCIBusId(out, (int)out_len, device));
  return 0;
}

static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static bool all_zero(const uint8_t *p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i]) return false; return true; }

int zkr_key_load_websnark(const void *pk_bin, size_t pk_len, int device, zkr_key **out) {
  if (!pk_bin || !out) { set_error("null argument"); return ZKR_ERR_ARG; }
  if (zkr_device_count() <= device || device < 0) { set_error("no HIP device %d (found %d); libzkr_hip has no CPU fallback", device, zkr_device_count()); return ZKR_ERR_NO_DEVICE; }
  const uint8_t *pk = (const uint8_t *)pk_bin;
  if (pk_len < 488) { set_error("proving key shorter than its fixed header (488 bytes)"); return ZKR_ERR_BAD_KEY; }
  uint32_t n = rd32(pk), p = rd32(pk + 4), m = rd32(pk + 8);
  uint32_t ptr[7];
  for (int i = 0; i < 7; i++) ptr[i] = rd32(pk + 12 + 4 * i);
  if (n == 0 || (uint64_t)p + 1 > n || m < 2 || (m & (m - 1))) { set_error("bad key geometry nVars=%u nPublic=%u domainSize=%u", n, p, m); return ZKR_ERR_BAD_KEY; }
  if (ptr[0] != 488) { set_error("polsA pointer %u != 488", ptr[0]); return ZKR_ERR_BAD_KEY; }
  // the two pols sections are scanned before anything else: their bounds must lie inside the caller's buffer
  if ((uint64_t)ptr[1] < 488 || (uint64_t)ptr[1] > (uint64_t)ptr[2] || (uint64_t)ptr[2] > (uint64_t)pk_len) {
    set_error("pols pointers out of order or past the buffer (polsB %u, pointsA %u, length %zu)", ptr[1], ptr[2], pk_len);
    return ZKR_ERR_BAD_KEY;
  }
  // point sections have fixed sizes (binarify.ts:129-138)
  uint64_t expect_end = (uint64_t)ptr[2] + 64ull * n + 64ull * n + 128ull * n + 64ull * (n - p - 1) + 64ull * m;
  if (ptr[3] != ptr[2] + 64ull * n || ptr[4] != ptr[3] + 64ull * n || ptr[5] != ptr[4] + 128ull * n || ptr[6] != ptr[5] + 64ull * (n - p - 1) ||
      expect_end != pk_len) {
    set_error("section pointers/length inconsistent with nVars=%u nPublic=%u domainSize=%u (len %zu)", n, p, m, pk_len);
    return ZKR_ERR_BAD_KEY;
  }
  // column-major sparse pols -> CSR rows
  std::vector<uint32_t> rowptr[2], col[2];
  std::vector<uint8_t> coef[2];
  for (int s = 0; s < 2; s++) {
    size_t o = ptr[s], end = ptr[s + 1];
    std::vector<uint32_t> cnt(m + 1, 0);
    size_t nnz = 0;
    for (uint32_t sig = 0; sig < n; sig++) {
      if (o + 4 > end) { set_error("pols section %d truncated", s); return ZKR_ERR_BAD_KEY; }
      uint32_t kk = rd32(pk + o);
      o += 4;
      if (o + 36ull * kk > end) { set_error("pols section %d truncated", s); return ZKR_ERR_BAD_KEY; }
      for (uint32_t e = 0; e < kk; e++, o += 36) {
        uint32_t c = rd32(pk + o);
        if (c >= m) { set_error("constraint index %u >= domainSize %u", c, m); return ZKR_ERR_BAD_KEY; }
        cnt[c + 1]++;
      }
      nnz += kk;
    }
    if (o != end) { set_error("pols section %d has trailing bytes", s); return ZKR_ERR_BAD_KEY; }
    rowptr[s].assign(m + 1, 0);
    for (uint32_t c = 0; c < m; c++) rowptr[s][c + 1] = rowptr[s][c] + cnt[c + 1];
    col[s].resize(nnz);
    coef[s].resize(nnz * 32);
    std::vector<uint32_t> cur(rowptr[s].begin(), rowptr[s].end() - 1);
    o = ptr[s];
    for (uint32_t sig = 0; sig < n; sig++) {
      uint32_t kk = rd32(pk + o);
      o += 4;
      for (uint32_t e = 0; e < kk; e++, o += 36) {
        uint32_t c = rd32(pk + o), pos = cur[c]++;
        col[s][pos] = sig;
        memcpy(&coef[s][(size_t)pos * 32], pk + o + 4, 32);
      }
    }
  }
  // point tables: drop points at infinity (x == 0), remember which scalar each kept point multiplies
  const void *src[N_TABLES] = {pk + ptr[2], pk + ptr[3], pk + ptr[4], pk + ptr[5], pk + ptr[6]};
  const bool on_dev[N_TABLES] = {false, false, false, false, false};
  std::vector<uint32_t> srcidx[N_TABLES], sidx[N_TABLES];
  for (uint32_t i = 0; i < n; i++) {
    if (!all_zero(pk + ptr[2] + 64ull * i, 32)) { srcidx[T_A].push_back(i); sidx[T_A].push_back(i); }
    if (!all_zero(pk + ptr[3] + 64ull * i, 32)) { srcidx[T_B1].push_back(i); sidx[T_B1].push_back(i); }
    if (!all_zero(pk + ptr[4] + 128ull * i, 64)) { srcidx[T_B2].push_back(i); sidx[T_B2].push_back(i); }
  }
  for (uint32_t i = 0; i + p + 1 < n; i++)
    if (!all_zero(pk + ptr[5] + 64ull * i, 32)) { srcidx[T_C].push_back(i); sidx[T_C].push_back(i + p + 1); }
  uint32_t logm = 0;
  while ((1u << logm) < m) logm++;
  for (uint32_t j = 0; j < m; j++) {  // h is produced in bit-reversed order: pair position j with hExps[bitrev(j)]
    uint32_t i = 0;
    for (uint32_t b = 0; b < logm; b++) i |= ((j >> b) & 1) << (logm - 1 - b);
    if (!all_zero(pk + ptr[6] + 64ull * i, 32)) { srcidx[T_H].push_back(i); sidx[T_H].push_back(j); }
  }
  return key_build(device, n, p, m, rowptr, col, coef, src, on_dev, srcidx, sidx, pk + 40, out);
}

void zkr_key_free(zkr_key *k) {
  if (!k) return;
  hipSetDevice(k->device);
  // the streams are the device's (one set per device, shared by its keys): wait for THIS key's work only -- the last events of its
  // proof slots and staged uploads (an event that was never recorded is complete) -- not for the proofs other keys have in flight
  for (ProofSlot &sl : k->slot) {
    for (auto e : sl.ev_red)
      if (e) hipEventSynchronize(e);
    for (int t = 0; t < N_TABLES; t++)
      if (sl.ev_res[t]) hipEventSynchronize(sl.ev_res[t]);
  }
  for (WitnessStage &ws : k->stage)
    if (ws.ev_up) hipEventSynchronize(ws.ev_up);
  for (ProofSlot &sl : k->slot) {
    for (int t = 0; t < N_TABLES; t++) {
      if (sl.ev_done[t]) hipEventDestroy(sl.ev_done[t]);
      if (sl.ev_sorted[t]) hipEventDestroy(sl.ev_sorted[t]);
      if (sl.ev_res[t]) hipEventDestroy(sl.ev_res[t]);
      msm_ws_free(sl.ws[t]);
    }
    if (sl.ev_w) hipEventDestroy(sl.ev_w);
    if (sl.ev_h) hipEventDestroy(sl.ev_h);
    for (auto e : sl.ev_red)
      if (e) hipEventDestroy(e);
    digit_lists_free(sl.dig_w); digit_lists_free(sl.dig_h);
    hipFree(sl.d_w); hipFree(sl.va); hipFree(sl.vb); hipFree(sl.ca); hipFree(sl.cb); hipFree(sl.d_h);
    for (auto e : sl.event_pool) hipEventDestroy(e);
  }
  for (WitnessStage &ws : k->stage) {
    if (ws.d_w) hipFree(ws.d_w);
    if (ws.ev_up) hipEventDestroy(ws.ev_up);
  }
  hipFree(k->tw29);
  hipFree(k->twl29);
  if (k->owns_arena) hipFree(k->arena);
  if (k->base_arena) hipFree(k->base_arena);
  delete k;
}

int zkr_key_info(const zkr_key *k, uint64_t out[10]) {
  if (!k || !out) { set_error("null argument"); return ZKR_ERR_ARG; }
  out[0] = k->h.n; out[1] = k->h.p; out[2] = k->h.m; out[3] = k->h.nnzA; out[4] = k->h.nnzB;
  for (int t = 0; t < N_TABLES; t++) out[5 + t] = k->h.npts[t];
  return 0;
}

int zkr_key_slots(const zkr_key *k) { return k ? PROOF_SLOTS : 0; }
int zkr_key_fuse(const zkr_key *k) { return k ? k->slot[0].cap : 0; }

int zkr_key_windows(const zkr_key *k, uint32_t c_out[5], uint32_t k_out[5]) {
  if (!k || !c_out || !k_out) { set_error("null argument"); return ZKR_ERR_ARG; }
  for (int t = 0; t < N_TABLES; t++) { c_out[t] = (uint32_t)k->plan[t].c; k_out[t] = (uint32_t)k->plan[t].K; }
  return 0;
}

int zkr_key_arena(const zkr_key *k, void **dev_ptr, size_t *len) {
  if (!k || !dev_ptr || !len) { set_error("null argument"); return ZKR_ERR_ARG; }
  *dev_ptr = k->arena;
  *len = k->arena_len;
  return 0;
}

int zkr_key_adopt_arena(void *dev_ptr, size_t len, int device, zkr_key **out) {
  if (!dev_ptr || !out) { set_error("null argument"); return ZKR_ERR_ARG; }
  if (zkr_device_count() <= device || device < 0) { set_error("no HIP device %d", device); return ZKR_ERR_NO_DEVICE; }
  ZKR_HIP_CHECK(hipSetDevice(device));
  ArenaHeader h;
  if (len < ARENA_HEADER_BYTES) { set_error("arena too small"); return ZKR_ERR_BAD_KEY; }
  ZKR_HIP_CHECK(hipMemcpy(&h, dev_ptr, sizeof(h), hipMemcpyDeviceToHost));
  if (const char *fault = arena_header_fault(h, len)) { set_error("arena header: %s", fault); return ZKR_ERR_BAD_KEY; }
  if (int rc = arena_rows_check((const unsigned char *)dev_ptr, h)) return rc;
  zkr_key *k = new zkr_key();
  k->device = device;
  k->arena = (unsigned char *)dev_ptr;
  k->arena_len = len;
  k->owns_arena = false;
  k->h = h;
  int rc = key_alloc_workspace(k);
  if (rc) { zkr_key_free(k); return rc; }
  *out = k;
  return 0;
}

// ---- compact form of the arena for replication over slow links: header + CSR rows + LEVEL 0 of every
// point table + rank maps; the window levels 1..K-1 (12/13 of the arena) and the twiddle tables are rebuilt by the
// receiver (msm_precompute_kernel, twiddle_table_kernel).  Same ArenaHeader with BASE_MAGIC and offsets of the compact
// layout; win_c / npts / share flags carry over so the rebuilt arena is byte-identical to the sender's.
constexpr uint64_t BASE_MAGIC = 0x32304245534b525aull;  // "ZRKSEB02"-ish tag: distinct from ARENA_MAGIC

}  // extern "C"
namespace zkr {
void base_layout(const ArenaHeader &full, ArenaHeader &b) {
  b = full;
  b.magic = BASE_MAGIC;
  size_t off = ARENA_HEADER_BYTES;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return (uint64_t)o; };
  b.off_tw = b.off_twl = 0;
  for (int s = 0; s < 2; s++) {
    uint32_t nnz = s == 0 ? full.nnzA : full.nnzB;
    b.off_rowptr[s] = take(((size_t)full.m + 1) * 4);
    b.off_col[s] = take((size_t)nnz * 4 + 4);
    b.off_coef[s] = take((size_t)nnz * 32 + 32);
    b.off_wide[s] = take((size_t)full.n_wide[s] * 4 + 4);
  }
  for (int t = 0; t < N_TABLES; t++) {
    size_t pb = t == T_B2 ? 128 : 64;
    b.off_pts[t] = take((size_t)full.npts[t] * pb + pb);
    b.off_rank[t] = take((size_t)rank_entries(full, t) * 4 + 4);
  }
  b.total_len = off;
}
}  // namespace zkr
extern "C" {

int zkr_key_base_arena(zkr_key *k, void **dev_ptr, size_t *len) {
  if (!k || !dev_ptr || !len) { set_error("null argument"); return ZKR_ERR_ARG; }
  ZKR_HIP_CHECK(hipSetDevice(k->device));
  std::lock_guard<std::mutex> lk(k->mu);
  if (!k->base_arena) {
    ArenaHeader b;
    base_layout(k->h, b);
    unsigned char *buf = nullptr;
    ZKR_HIP_CHECK(hipMalloc(&buf, b.total_len));
    auto cp = [&](uint64_t dst, uint64_t src, size_t bytes) { return bytes ? hipMemcpy(buf + dst, k->arena + src, bytes, hipMemcpyDeviceToDevice) : hipSuccess; };
    hipError_t e = hipMemset(buf, 0, b.total_len);
    if (e == hipSuccess) e = hipMemcpy(buf, &b, sizeof(b), hipMemcpyHostToDevice);
    for (int s = 0; s < 2 && e == hipSuccess; s++) {
      uint32_t nnz = s == 0 ? k->h.nnzA : k->h.nnzB;
      e = cp(b.off_rowptr[s], k->h.off_rowptr[s], ((size_t)k->h.m + 1) * 4);
      if (e == hipSuccess) e = cp(b.off_col[s], k->h.off_col[s], (size_t)nnz * 4);
      if (e == hipSuccess) e = cp(b.off_coef[s], k->h.off_coef[s], (size_t)nnz * 32);
      if (e == hipSuccess) e = cp(b.off_wide[s], k->h.off_wide[s], (size_t)k->h.n_wide[s] * 4);
    }
    for (int t = 0; t < N_TABLES && e == hipSuccess; t++) {
      e = cp(b.off_pts[t], k->h.off_pts[t], (size_t)k->h.npts[t] * (t == T_B2 ? 128 : 64));
      if (e == hipSuccess) e = cp(b.off_rank[t], k->h.off_rank[t], (size_t)rank_entries(k->h, t) * 4);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { hipFree(buf); set_error("building the compact arena failed: %s", hipGetErrorString(e)); return ZKR_ERR_HIP; }
    // the compact form carries the base points in the key's own radix (2^256): the receiver rebuilds the window levels from
    // them with msm_precompute, exactly as a key load does
    for (int t = 0; t < N_TABLES; t++) {
      int rc = radix_convert(k->device, t == T_B2, buf + b.off_pts[t], k->h.npts[t], false);
      if (rc) { hipFree(buf); return rc; }
    }
    k->base_arena = buf;
    k->base_arena_len = b.total_len;
  }
  *dev_ptr = k->base_arena;
  *len = k->base_arena_len;
  return 0;
}

int zkr_key_adopt_base_arena(const void *dev_ptr, size_t len, int device, zkr_key **out) {
  if (!dev_ptr || !out) { set_error("null argument"); return ZKR_ERR_ARG; }
  if (zkr_device_count() <= device || device < 0) { set_error("no HIP device %d", device); return ZKR_ERR_NO_DEVICE; }
  ZKR_HIP_CHECK(hipSetDevice(device));
  if (len < ARENA_HEADER_BYTES) { set_error("compact arena too small"); return ZKR_ERR_BAD_KEY; }
  ArenaHeader b;
  ZKR_HIP_CHECK(hipMemcpy(&b, dev_ptr, sizeof(b), hipMemcpyDeviceToHost));
  if (b.magic != BASE_MAGIC || b.total_len != len) { set_error("compact arena header mismatch (magic/len)"); return ZKR_ERR_BAD_KEY; }
  // the full layout, exactly as key_build lays it out
  ArenaHeader h = b;
  h.magic = ARENA_MAGIC;
  MsmPlan plan[N_TABLES];
  for (int t = 0; t < N_TABLES; t++) {
    if (h.win_c[t] < 2 || h.win_c[t] > 26) { set_error("compact arena: bad window size"); return ZKR_ERR_BAD_KEY; }
    plan[t] = msm_plan(rank_entries(h, t), h.npts[t], (int)h.win_c[t]);
  }
  if (h.sc_n[0] > h.n || h.sc_lo[0] > h.n - h.sc_n[0] || h.sc_n[1] > h.m || h.sc_lo[1] > h.m - h.sc_n[1]) { set_error("compact arena: scalar ranges outside the vectors"); return ZKR_ERR_BAD_KEY; }
  arena_layout(h);
  const size_t off = h.total_len;
  // every section of the compact form lies inside the bytes handed over (a truncated or edited header must not drive the copies)
  {
    if (h.logm > 27 || h.m != (1u << h.logm) || h.n == 0 || h.p >= h.n) { set_error("compact arena: inconsistent sizes"); return ZKR_ERR_BAD_KEY; }
    auto inside = [&](uint64_t o, uint64_t bytes) { return o >= ARENA_HEADER_BYTES && o <= len && bytes <= len - o; };
    bool ok = true;
    for (int s = 0; s < 2 && ok; s++) {
      const uint64_t nnz = s == 0 ? h.nnzA : h.nnzB;
      ok = h.n_wide[s] <= h.m && inside(b.off_rowptr[s], ((uint64_t)h.m + 1) * 4) && inside(b.off_col[s], nnz * 4) && inside(b.off_coef[s], nnz * 32) &&
           inside(b.off_wide[s], (uint64_t)h.n_wide[s] * 4);
    }
    for (int t = 0; t < N_TABLES && ok; t++) {
      const uint64_t nsc = rank_entries(h, t);
      ok = h.npts[t] <= nsc && inside(b.off_pts[t], (uint64_t)h.npts[t] * (t == T_B2 ? 128 : 64)) && inside(b.off_rank[t], nsc * 4);
    }
    if (!ok) { set_error("compact arena: a section lies outside the %zu bytes handed over", len); return ZKR_ERR_BAD_KEY; }
  }
  unsigned char *arena = nullptr;
  ZKR_HIP_CHECK(hipMalloc(&arena, off));
  if (hipMemset(arena, 0, off) != hipSuccess) { hipFree(arena); set_error("hipMemset failed"); return ZKR_ERR_HIP; }  // as key_build: gaps and slack are defined bytes
  const unsigned char *src = (const unsigned char *)dev_ptr;
  auto cp = [&](uint64_t dst, uint64_t from, size_t bytes) { return bytes ? hipMemcpy(arena + dst, src + from, bytes, hipMemcpyDeviceToDevice) : hipSuccess; };
  hipError_t e = hipMemcpy(arena, &h, sizeof(h), hipMemcpyHostToDevice);
  for (int s = 0; s < 2 && e == hipSuccess; s++) {
    uint32_t nnz = s == 0 ? h.nnzA : h.nnzB;
    e = cp(h.off_rowptr[s], b.off_rowptr[s], ((size_t)h.m + 1) * 4);
    if (e == hipSuccess) e = cp(h.off_col[s], b.off_col[s], (size_t)nnz * 4);
    if (e == hipSuccess) e = cp(h.off_coef[s], b.off_coef[s], (size_t)nnz * 32);
    if (e == hipSuccess) e = cp(h.off_wide[s], b.off_wide[s], (size_t)h.n_wide[s] * 4);
  }
  for (int t = 0; t < N_TABLES && e == hipSuccess; t++) {
    e = cp(h.off_pts[t], b.off_pts[t], (size_t)h.npts[t] * (t == T_B2 ? 128 : 64));
    if (e == hipSuccess) e = cp(h.off_rank[t], b.off_rank[t], (size_t)rank_entries(h, t) * 4);
  }
  if (e != hipSuccess) { hipFree(arena); set_error("unpacking the compact arena failed: %s", hipGetErrorString(e)); return ZKR_ERR_HIP; }
  for (int t = 0; t < N_TABLES; t++) {
    int rc = msm_precompute(device, t == T_B2, arena + h.off_pts[t], h.npts[t], plan[t]);
    if (rc) { hipFree(arena); return rc; }
  }
  twiddle_table_kernel<<<(h.m + 255) / 256, 256>>>((Fr *)(arena + h.off_tw), h.m, fr_root_of_unity(h.logm + 1));
  twiddle_table_kernel<<<((1u << TWL_LOG) + 255) / 256, 256>>>((Fr *)(arena + h.off_twl), 1u << TWL_LOG, fr_root_of_unity(TWL_LOG + 1));
  e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { hipFree(arena); set_error("rebuilding the window tables failed: %s", hipGetErrorString(e)); return ZKR_ERR_HIP; }
  zkr_key *k = new zkr_key();
  k->device = device;
  k->arena = arena;
  k->arena_len = off;
  k->owns_arena = true;
  k->h = h;
  int rc = key_alloc_workspace(k);
  if (rc) { zkr_key_free(k); return rc; }
  *out = k;
  return 0;
}

int zkr_key_save(const zkr_key *k, const char *path) {
  if (!k || !path) { set_error("null argument"); return ZKR_ERR_ARG; }
  ZKR_HIP_CHECK(hipSetDevice(k->device));
  FILE *f = fopen(path, "wb");
  if (!f) { set_error("cannot open %s for writing", path); return ZKR_ERR_ARG; }
  const size_t CHUNK = (size_t)64 << 20;
  void *stage = nullptr;
  hipError_t e = hipHostMalloc(&stage, CHUNK, hipHostMallocDefault);
  int rc = 0;
  if (e != hipSuccess) { set_error("pinned staging buffer: %s", hipGetErrorString(e)); rc = ZKR_ERR_HIP; }
  for (size_t off = 0; !rc && off < k->arena_len; off += CHUNK) {
    size_t nb = k->arena_len - off < CHUNK ? k->arena_len - off : CHUNK;
    e = hipMemcpy(stage, k->arena + off, nb, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("arena download failed: %s", hipGetErrorString(e)); rc = ZKR_ERR_HIP; }
    else if (fwrite(stage, 1, nb, f) != nb) { set_error("short write to %s", path); rc = ZKR_ERR_ARG; }
  }
  if (stage) hipHostFree(stage);
  if (fclose(f) != 0 && !rc) { set_error("close failed on %s", path); rc = ZKR_ERR_ARG; }
  return rc;
}

int zkr_key_load_file(const char *path, int device, zkr_key **out) {
  if (!path || !out) { set_error("null argument"); return ZKR_ERR_ARG; }
  if (zkr_device_count() <= device || device < 0) { set_error("no HIP device %d (found %d); libzkr_hip has no CPU fallback", device, zkr_device_count()); return ZKR_ERR_NO_DEVICE; }
  ZKR_HIP_CHECK(hipSetDevice(device));
  FILE *f = fopen(path, "rb");
  if (!f) { set_error("cannot open %s", path); return ZKR_ERR_BAD_KEY; }
  ArenaHeader h;
  if (fread(&h, 1, sizeof(h), f) != sizeof(h) || h.magic != ARENA_MAGIC || h.total_len < ARENA_HEADER_BYTES) {
    fclose(f);
    set_error("%s is not a packed zkr key (bad header)", path);
    return ZKR_ERR_BAD_KEY;
  }
  fseek(f, 0, SEEK_END);
  long flen = ftell(f);
  if (flen < 0 || (uint64_t)flen != h.total_len) { fclose(f); set_error("%s: length %ld != header total %llu", path, flen, (unsigned long long)h.total_len); return ZKR_ERR_BAD_KEY; }
  if (const char *fault = arena_header_fault(h, h.total_len)) { fclose(f); set_error("%s: %s", path, fault); return ZKR_ERR_BAD_KEY; }
  fseek(f, 0, SEEK_SET);
  unsigned char *arena = nullptr;
  const size_t CHUNK = (size_t)64 << 20;
  void *stage = nullptr;
  int rc = 0;
  hipError_t e = hipMalloc(&arena, h.total_len);
  if (e == hipSuccess) e = hipHostMalloc(&stage, CHUNK, hipHostMallocDefault);
  if (e != hipSuccess) { set_error("allocation failed: %s", hipGetErrorString(e)); rc = ZKR_ERR_HIP; }
  for (size_t off = 0; !rc && off < h.total_len; off += CHUNK) {
    size_t nb = h.total_len - off < CHUNK ? h.total_len - off : CHUNK;
    if (fread(stage, 1, nb, f) != nb) { set_error("short read from %s", path); rc = ZKR_ERR_BAD_KEY; }
    else if ((e = hipMemcpy(arena + off, stage, nb, hipMemcpyHostToDevice)) != hipSuccess) { set_error("arena upload failed: %s", hipGetErrorString(e)); rc = ZKR_ERR_HIP; }
  }
  fclose(f);
  if (stage) hipHostFree(stage);
  if (!rc) rc = arena_rows_check(arena, h);
  if (rc) { hipFree(arena); return rc; }
  zkr_key *k = new zkr_key();
  k->device = device;
  k->arena = arena;
  k->arena_len = h.total_len;
  k->owns_arena = true;
  k->h = h;
  rc = key_alloc_workspace(k);
  if (rc) { zkr_key_free(k); return rc; }
  *out = k;
  return 0;
}

void zkr_free(void *p) { free(p); }

}  // extern "C"
