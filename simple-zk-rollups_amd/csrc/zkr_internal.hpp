// zkr_internal.hpp -- host-side structures shared by the translation units of libzkr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <time.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>
#include "../../include/zkr.h"
#include "hostops.hpp"
#include "shard_group.hpp"

namespace zkr {

struct Tw29;  // kernels_ntt.hpp: a butterfly twiddle as nine 29-bit limbs
void set_error(const char *fmt, ...);
#define ZKR_HIP_CHECK(expr)                                                                 \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ZKR_ERR_HIP;                                                                   \
    }                                                                                       \
  } while (0)

// a device allocation (or a HIP event) that goes when its scope ends: the test hooks and one-off entry points return early on
// every failed HIP call, and none of those paths may leave device memory behind
struct DevBuf {
  void *p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) hipFree(p); }
  int alloc(size_t bytes) { ZKR_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1)); return 0; }
  template <class T> T *as() const { return static_cast<T *>(p); }
  void *release() { void *q = p; p = nullptr; return q; }
};
struct ScopedEvent {
  hipEvent_t e = nullptr;
  ScopedEvent() = default;
  ScopedEvent(const ScopedEvent &) = delete;
  ScopedEvent &operator=(const ScopedEvent &) = delete;
  ~ScopedEvent() { if (e) hipEventDestroy(e); }
  int create() { ZKR_HIP_CHECK(hipEventCreate(&e)); return 0; }
};

enum { T_A = 0, T_B1 = 1, T_B2 = 2, T_C = 3, T_H = 4, N_TABLES = 5 };

// The device key is ONE position-independent arena: header + sections addressed by byte offsets,
// so a replica on another GPU is a single broadcast of [arena, arena+len).
struct ArenaHeader {
  uint64_t magic;      // "ZKRKEY04"
  uint64_t total_len;
  uint32_t n, p, m, logm;
  uint32_t nnzA, nnzB;
  uint32_t npts[N_TABLES];
  uint32_t tlog;       // twiddle table entries = 2^tlog (= m)
  uint64_t off_tw, off_twl;
  uint64_t off_rowptr[2], off_col[2], off_coef[2];
  uint64_t off_wide[2];  // row indices of the QAP rows wider than SPMV_WIDE terms (spmv_wide_kernel)
  uint32_t n_wide[2];
  uint64_t off_pts[N_TABLES], off_rank[N_TABLES];  // rank: u32 per scalar of the table's vector -> index of its point, RANK_NONE if dropped
  uint8_t alfa1[64], beta1[64], delta1[64];  // Montgomery affine, as in the websnark key header
  uint8_t beta2[128], delta2[128];
  uint32_t share_b;    // B1 and B2 keep the same signals (always true for honest keys): one digit sort serves both
  uint32_t win_c[N_TABLES];  // window bits of each table: off_pts[t] holds K = ceil(255/c) x npts[t] points, level k = 2^(ck) * base
  uint32_t share_ac;   // A and C are laid out over the union of their supports (missing points stored as infinity): one digit sort serves both
  uint32_t rank_identity[N_TABLES];  // rank[s] == s for every scalar of the table's vector: the sort skips the gather
  // Intra-proof sharding (zkr_key_shard): a shard key multiplies only the scalars [sc_lo, sc_lo + sc_n) of
  // each scalar vector (index 0: the witness w, serving A, B1, B2, C; index 1: h, serving H) and holds only their points; its rank
  // maps have sc_n entries (rank[i] = point of scalar sc_lo + i).  A whole key: sc_lo = 0, sc_n = {n, m}, shard_parts = 1.
  uint32_t sc_lo[2], sc_n[2];
  uint32_t shard_part, shard_parts;
};
static_assert(sizeof(ArenaHeader) <= 1024, "header fits its slot");
constexpr size_t ARENA_HEADER_BYTES = 1024;
constexpr uint64_t ARENA_MAGIC = 0x343059454b524b5aull;  // "ZKRKEY04" (03: point tables in the radix-2^261 form of field29.hpp; 04: scalar sub-ranges of shard keys): bump with every change of ArenaHeader or of a section layout (packed key files carry it)
inline uint32_t rank_entries(const ArenaHeader &h, int t) { return h.sc_n[t == T_H ? 1 : 0]; }  // entries of table t's rank map = scalars the key multiplies with it

// digit records of one scalar vector, split by bucket range (kernels_msm.hpp "digit sort", stage 1)
struct DigitLists {
  uint32_t *rng = nullptr;  // counts[range][XCD slot] | fill cursors[range][XCD slot] | range offsets[nR + 1]  (kernels_msm.hpp DIGIT_RNG_WORDS)
  uint32_t *ent_s = nullptr, *ent_b = nullptr;
};

struct MsmWorkspace {
  uint32_t *counts = nullptr, *offsets = nullptr, *entries = nullptr;
  uint32_t *size_hist = nullptr, *order = nullptr;  // [2][SIZE_BINS] size-class histogram + hand-out counters; bucket ids fullest first
  uint32_t *chunk_cnt = nullptr;  // [K][J][nbw] per-chunk bucket occupancies, then per-bucket prefixes over chunks
  DigitLists own_dig;             // digit records when the workspace is not attached to a key (stage hooks)
  uint32_t *big_list = nullptr, *big_count = nullptr, *block_sums = nullptr;
  void *big_partials = nullptr;
  void *buckets = nullptr, *group_out = nullptr, *task_out = nullptr, *result = nullptr;
  void *h_result = nullptr;  // pinned host copy of the MSM result point (XYZZ)
  size_t max_nb = 0, max_entries = 0;
  size_t sets = 1;  // bucket sets (and reduction buffers) per proof: 2 in B1's workspace when A is reduced with it (zkr_key.hip alloc_msm_ws)
};

struct ProfStage {
  std::string name;
  double ms = 0;
  uint64_t launches = 0;
};
struct ProfSpan {
  int stage;
  hipEvent_t e0, e1;
};

constexpr int LAT_GLOG = 3;  // reduction groups of a chain nothing can hide (a synchronous proof's last one): 2^3 buckets, see msm_reduce_enqueue
struct MsmPlan {
  int c, K, glog;
  uint32_t nbw, nb, big_thresh;  // nb = nbw = 2^(c-1) buckets, one set shared by the K windows
  uint32_t nR, nbl;   // digit sort: nR bucket ranges of nbl buckets (LDS counters of one workgroup)
  uint32_t J;         // digit sort: J chunks per bucket range
  uint32_t S;         // reduction: workgroups per task in msm_reduce2_kernel
};
// two tables whose bucket sets ONE reduction launch set can walk end to end (msm_reduce*_kernel take `batch` sets of one geometry)
inline bool same_reduce_geometry(const MsmPlan &a, const MsmPlan &b) { return a.c == b.c && a.K == b.K && a.nbw == b.nbw && a.nb == b.nb && a.glog == b.glog && a.S == b.S; }

}  // namespace zkr

namespace zkr {
// Everything one proof in flight owns: witness + calcH vectors, digit codes, the five MSM workspaces, its
// events and timing spans.  A key has PROOF_SLOTS of them so that the GPU work of the next proof is enqueued
// (zkr_prove_submit) while the host still assembles the previous one (zkr_prove_collect).
constexpr int PROOF_SLOTS = 2;  // three measured the same (134.1 against 134.1 / 134.3 proofs/s, HISTORY.md 7b)
constexpr int MAX_FUSE = 16;  // most proofs one launch set carries (zkr_key.hip fused_capacity)
// Host witnesses (zkr_prove, zkr_prove_batch: the ArrayBuffer of binarifyWitness) reach the GPU through a ring of device
// staging buffers, one more than there are proof slots: the witness of the NEXT proof crosses PCIe while both slots
// compute (and outside the key's lock), so a stream of host-buffer calls keeps the GPU as busy as device-resident
// witnesses do.
constexpr int STAGE_BUFS = PROOF_SLOTS + 1;
struct WitnessStage {
  Fr *d_w = nullptr;         // its device copy, read by ingest_kernel
  hipEvent_t ev_up = nullptr;
  bool busy = false;
};
struct ProofSlot {
  Fr *d_w = nullptr, *va = nullptr, *vb = nullptr, *ca = nullptr, *cb = nullptr, *d_h = nullptr;
  DigitLists dig_w, dig_h;  // digit records of w (shared by A, B1, B2, C) and of h
  MsmWorkspace ws[N_TABLES];
  hipEvent_t ev_w = nullptr, ev_h = nullptr;
  hipEvent_t ev_red[N_TABLES] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // end of the proof's work on each reduction stream
  hipEvent_t ev_done[N_TABLES] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_sorted[N_TABLES] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_res[N_TABLES] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // the table's result point has landed in its pinned host buffer
  bool res_pending[N_TABLES] = {false, false, false, false, false};
  int cap = 1;    // proofs one submit can fuse into shared launches (small circuits; every buffer above is cap times one proof's)
  int nbat = 0;   // proofs of the group in flight
  bool merged_ch = false;  // this group's H was accumulated onto C's bucket set: C's workspace holds C + H
  bool joint_ab = false;   // this group's A was accumulated behind B1's bucket sets and reduced with them in one chain: B1's workspace holds
                           // the results of both, B1's first (zkr_prove.hip prove_submit_enqueue)
  std::vector<uint8_t> rb, sb;  // blinding scalars of the proofs in flight, cap x 32 B each
  bool busy = false, collecting = false;
  std::vector<ProfSpan> spans;
  std::vector<hipEvent_t> event_pool;
  size_t event_next = 0;
};
}  // namespace zkr

struct zkr_key {
  int device = 0;
  unsigned char *arena = nullptr;
  size_t arena_len = 0;
  bool owns_arena = true;
  unsigned char *base_arena = nullptr;  // compact form for replication (zkr_key_base_arena), built on first request
  size_t base_arena_len = 0;
  zkr::ArenaHeader h;
  hipStream_t stream = nullptr;
  zkr::WitnessStage stage[zkr::STAGE_BUFS];
  std::mutex stage_mu;
  std::condition_variable stage_freed;
  hipStream_t prep_stream = nullptr;               // digit records, digit sorts, calcH
  hipStream_t red_stream[zkr::N_TABLES] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // reduction chains: [0] the G2 table, the G1 tables
                                                                                           // round-robin over [1..n_red)
  void *streams_owner = nullptr;                   // the DeviceStreams set (zkr_key.hip) the stream handles below come from
  std::mutex *enqueue_mu = nullptr;                // the device's enqueue lock (shared streams: one proof's launches are enqueued without interleaving)
  hipStream_t aux_stream = nullptr;                // = red_stream[n_all - 1]: C's oversized-bucket partial sums when C shares H's bucket set (zkr_prove.hip c_big)
  int n_red = 1;                                   // streams the reduction chains rotate over
  int n_all = 1;                                   // streams in red_stream[] (n_red + the auxiliary one)
  zkr::ProofSlot slot[zkr::PROOF_SLOTS];
  int next_slot = 0;
  std::mutex mu;  // slot hand-out, the enqueue phase of a proof (so two host threads do not interleave launches), stage totals
  std::shared_mutex split_mu;  // a shard key: held EXCLUSIVELY by the sharded proof that splits calcH over it and its siblings (zkr_multi.hip
                               // run_sharded), SHARED by a shard proving on its own (zkr_prove_partial outside a split group)
  std::atomic<int> split_checked{0};  // a shard key: 0 = the split calcH has not been compared with the replicated form on this shard set yet
                                      // (or the comparison could not be made: it is tried again), 1 = compared, identical, 2 = it DISAGREED:
                                      // never split again (zkr_multi.hip run_sharded)
  int replica_mode = 0;        // how the key came to its device: 0 = loaded / built there, ZKR_REPLICATE_FULL / _BASE = zkr_key_replicate in that form
  bool replica_direct = false;  // ... and whether the two devices could address each other
  std::condition_variable slot_freed;
  zkr::MsmPlan plan[zkr::N_TABLES];
  // proof assembly on the host: 4-bit window tables of delta_1 / delta_2 (built on first use)
  zkr::Tw29 *tw29 = nullptr, *twl29 = nullptr;  // butterfly twiddles, derived from the arena's tables when the key is set up (not part of the arena)
  std::once_flag delta_once;
  std::vector<zkr::G1Affine> delta1_tab;
  std::vector<zkr::G2Affine> delta2_tab;
  // profiling
  bool prof_on = false;
  std::vector<zkr::ProfStage> stages;
};

namespace zkr {
// zkr_key.hip
int key_build(int device, uint32_t n, uint32_t p, uint32_t m, const std::vector<uint32_t> rowptr[2], const std::vector<uint32_t> col[2],
              const std::vector<uint8_t> coef[2], const void *const tbl_src[N_TABLES], const bool tbl_src_on_device[N_TABLES],
              const std::vector<uint32_t> tbl_srcidx[N_TABLES], const std::vector<uint32_t> tbl_sidx[N_TABLES], const uint8_t *consts448,
              zkr_key **out);
int key_alloc_workspace(zkr_key *k);
int fixed_base_points(int device, bool g2, const uint8_t *scalars_std, size_t n, void **d_out, bool wipe_scalars = false);  // device array of affine Montgomery points
// zkr_prove.hip
struct Prof {  // where a launch helper records its timing spans (null key: stage hooks, no timing)
  zkr_key *k;
  ProofSlot *sl;
};
int prof_begin(Prof pf, hipStream_t s, const char *stage);
void prof_end(Prof pf, hipStream_t s, int span);
int prof_collect(zkr_key *k, ProofSlot &sl);
struct NttTables { const Fr *tw; const Tw29 *tw29, *twl29; int tlog; };  // x 2^256 powers of w_{2m} (coset factors); x 2^261 powers for the butterflies (kernels_ntt.hpp)
int ntt_lds_check(int device);  // ZKR_ERR_NO_DEVICE with a clear message when the device cannot hold an NTT tile in LDS
int ntt_tables29_build(const Fr *tw, uint32_t n_tw, const Fr *twl, uint32_t n_twl, hipStream_t s, Tw29 **tw29, Tw29 **twl29);
int run_ntt(hipStream_t s, const Fr *in0, const Fr *in1, Fr *out, const NttTables &tb, int L, bool dif, bool inverse, int pre, int nbat, Prof pf,
            const Fr *in0_b = nullptr, const Fr *in1_b = nullptr, Fr *out_b = nullptr,  // in0_b / in1_b / out_b: a second transform of the same shape in the same launches
            uint32_t want_lo = 0, uint32_t want_n = 0,   // DIF: only this range of the output is wanted (0, 0: all of it)
            uint32_t coset_shift = 0, uint32_t coset_add = 0);  // PRE_COSET when the transform is one block of a larger one (NttPassArgs::pre_shift)
int calc_h_device(zkr_key *k, ProofSlot &sl, hipStream_t s, int nbat = 1);  // sl.d_w -> sl.d_h (bit-reversed), nbat vectors end to end

// ShardGroup (the shards of one proof running concurrently; barrier + abort): shard_group.hpp, no HIP in it
// set by run_sharded's worker threads around zkr_prove_partial(_device): the proof this thread enqueues is part `shard_group_part`
// of that group (null: a shard proving on its own -- replicated calcH)
extern thread_local ShardGroup *shard_group;
extern thread_local unsigned shard_group_part;
extern thread_local bool shard_turn_held;  // the caller of this thread's zkr_prove_partial(_device) holds the shard's split_mu already (run_sharded with a split calcH)
int fused_capacity(const ArenaHeader &h, const MsmPlan plan[N_TABLES]);
void arena_layout(ArenaHeader &h);  // section offsets and total_len from the sizes in the header (n, p, m, nnz, n_wide, npts, win_c, sc_n): THE layout, whoever builds an arena
const char *arena_header_fault(const ArenaHeader &h, size_t len);  // null when a full arena's header is consistent with its sizes
void base_layout(const ArenaHeader &full, ArenaHeader &b);  // the same for the compact form (zkr_key_base_arena)
MsmPlan msm_plan(size_t n_scalars, size_t n_points, int c_fixed = 0);
uint32_t big_threshold(size_t n_points, int K, uint32_t nbw, int nbat);  // occupancy above which a bucket goes to msm_big_kernel (zkr_key.hip)
int digit_lists_alloc(DigitLists &dl, size_t n_scalars, const MsmPlan &pl);
void digit_lists_free(DigitLists &dl);
int msm_precompute(int device, bool g2, void *d_table, uint32_t n, const MsmPlan &pl);  // fills levels 1..K-1 of a table whose level 0 is in place, then converts the table to the hot path's radix
int radix_convert(int device, bool g2, void *d_points, size_t count, bool to261);
}  // namespace zkr
