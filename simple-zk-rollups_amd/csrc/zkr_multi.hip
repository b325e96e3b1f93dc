// zkr_multi.hip -- several GPUs of one node behind the C ABI, without torch.distributed: key replication device to
// device and a batch call that shards independent proofs over the replicas, one host thread per key.
//
// SURVEY.md 8(b) "Threading" (batch-of-64 driver: one host thread per GPU, hipSetDevice per worker) and 8(e): the rollup
// operator's host is Node (/root/reference/operator/src/snarks/common.ts:23-29, scripts/index.js:40-46) -- one process that
// awaits proofs -- so the proof-level sharding of BASELINE configs[3] has to be reachable from ONE process through the
// library the N-API shim binds, not only from one-process-per-GPU launchers (python/zkr_hip/batch.py keeps that form for
// torchrun + RCCL).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "zkr_internal.hpp"

using namespace zkr;

namespace {

constexpr size_t PEER_PIECE = (size_t)1 << 30;  // bytes per peer copy (arenas reach 78 GB at 2^24)
constexpr uint32_t RANK_NONE_U32 = 0xffffffffu;  // kernels_msm.hpp RANK_NONE (this unit includes no kernel header)

// [src, src + len) on src_dev -> dst on dst_dev, in pieces; same device: a plain device copy
int copy_between_devices(void *dst, int dst_dev, const void *src, int src_dev, size_t len) {
  ZKR_HIP_CHECK(hipSetDevice(dst_dev));
  for (size_t off = 0; off < len; off += PEER_PIECE) {
    const size_t nb = len - off < PEER_PIECE ? len - off : PEER_PIECE;
    if (dst_dev == src_dev) ZKR_HIP_CHECK(hipMemcpyAsync((char *)dst + off, (const char *)src + off, nb, hipMemcpyDeviceToDevice, nullptr));
    else ZKR_HIP_CHECK(hipMemcpyPeerAsync((char *)dst + off, dst_dev, (const char *)src + off, src_dev, nb, nullptr));
  }
  ZKR_HIP_CHECK(hipStreamSynchronize(nullptr));
  return 0;
}

bool peer_direct_ask(int dst_dev, int src_dev);
// direct loads/stores between the two devices (xGMI on an MI355X node).  Without it hipMemcpyPeer still works -- staged
// through host memory by the runtime -- which is what the compact form is for.
bool peer_direct(int dst_dev, int src_dev) {
  if (dst_dev == src_dev) return true;
  static std::atomic<int> known[16][16];  // 0 not asked yet, 1 yes, 2 no: asked once per pair (a sharded proof asks for every pair)
  const bool cached = dst_dev >= 0 && dst_dev < 16 && src_dev >= 0 && src_dev < 16;
  if (cached && known[dst_dev][src_dev].load()) return known[dst_dev][src_dev].load() == 1;
  const bool ok = peer_direct_ask(dst_dev, src_dev);
  if (cached) known[dst_dev][src_dev].store(ok ? 1 : 2);
  return ok;
}
bool peer_direct_ask(int dst_dev, int src_dev) {
  int can = 0;
  if (hipDeviceCanAccessPeer(&can, dst_dev, src_dev) != hipSuccess || !can) return false;
  if (hipSetDevice(dst_dev) != hipSuccess) return false;
  hipError_t e = hipDeviceEnablePeerAccess(src_dev, 0);
  if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
  return e == hipSuccess;
}

}  // namespace

extern "C" {

int zkr_key_replicate(const zkr_key *src, int dst_device, int mode, zkr_key **out) {
  if (!src || !out) { set_error("null argument"); return ZKR_ERR_ARG; }
  if (mode != ZKR_REPLICATE_AUTO && mode != ZKR_REPLICATE_FULL && mode != ZKR_REPLICATE_BASE) { set_error("unknown replication mode %d", mode); return ZKR_ERR_ARG; }
  if (zkr_device_count() <= dst_device || dst_device < 0) { set_error("no HIP device %d (found %d)", dst_device, zkr_device_count()); return ZKR_ERR_NO_DEVICE; }
  if (const char *e = getenv("ZKR_REPLICATE_MODE")) {  // experiments / tests: force a form whatever the caller asked for
    if (!strcmp(e, "full")) mode = ZKR_REPLICATE_FULL;
    else if (!strcmp(e, "base")) mode = ZKR_REPLICATE_BASE;
  }
  const bool direct = peer_direct(dst_device, src->device);
  if (mode == ZKR_REPLICATE_AUTO) mode = direct ? ZKR_REPLICATE_FULL : ZKR_REPLICATE_BASE;
  if (mode == ZKR_REPLICATE_FULL) {
    // the whole arena, window tables included: nothing is recomputed on the receiver (4.47 GB at 2^20: ~30 ms per xGMI link)
    ZKR_HIP_CHECK(hipSetDevice(dst_device));
    DevBuf buf;
    if (int rc = buf.alloc(src->arena_len)) return rc;
    if (int rc = copy_between_devices(buf.p, dst_device, src->arena, src->device, src->arena_len)) return rc;
    zkr_key *k = nullptr;
    int rc = zkr_key_adopt_arena(buf.p, src->arena_len, dst_device, &k);
    if (rc) return rc;
    k->owns_arena = true;  // unlike an adopted broadcast buffer, this copy belongs to the replica
    k->replica_mode = ZKR_REPLICATE_FULL; k->replica_direct = direct;
    buf.release();
    *out = k;
    return 0;
  }
  // compact form: base points + QAP rows (1/10 of the bytes), window levels and twiddles rebuilt on the receiver
  void *base = nullptr;
  size_t base_len = 0;
  if (int rc = zkr_key_base_arena(const_cast<zkr_key *>(src), &base, &base_len)) return rc;
  int rc;
  if (dst_device == src->device) rc = zkr_key_adopt_base_arena(base, base_len, dst_device, out);
  else {
    ZKR_HIP_CHECK(hipSetDevice(dst_device));
    DevBuf tmp;
    if ((rc = tmp.alloc(base_len))) return rc;
    if ((rc = copy_between_devices(tmp.p, dst_device, base, src->device, base_len))) return rc;
    rc = zkr_key_adopt_base_arena(tmp.p, base_len, dst_device, out);
  }
  if (!rc) { (*out)->replica_mode = ZKR_REPLICATE_BASE; (*out)->replica_direct = direct; }
  return rc;
}

}  // extern "C"

namespace {

// proofs i = j, j + n_keys, j + 2 n_keys, ... of a batch on key j: gathered arguments, scattered results
struct Share {
  std::vector<const void *> wits;
  std::vector<uint8_t> rs, ss, proofs;
  int rc = 0;
  std::string err;
};

template <class Call>
int run_shares(zkr_key *const *keys, size_t n_keys, const void *const *witnesses, size_t count, const uint8_t *r32s, const uint8_t *s32s,
               uint8_t *proofs_out, Call call) {
  if (!keys || n_keys == 0 || (!witnesses && count) || !proofs_out) { set_error("null argument"); return ZKR_ERR_ARG; }
  if ((r32s == nullptr) != (s32s == nullptr)) { set_error("pass both r and s or neither"); return ZKR_ERR_ARG; }
  for (size_t j = 0; j < n_keys; j++) {
    if (!keys[j]) { set_error("key %zu is null", j); return ZKR_ERR_ARG; }
    if (keys[j]->h.n != keys[0]->h.n || keys[j]->h.m != keys[0]->h.m || keys[j]->h.p != keys[0]->h.p || memcmp(keys[j]->h.delta1, keys[0]->h.delta1, 64)) {
      set_error("key %zu is not a replica of key 0 (different circuit or setup)", j);
      return ZKR_ERR_ARG;
    }
    for (size_t i = 0; i < j; i++)
      if (keys[i] == keys[j]) { set_error("key %zu is listed twice: replicate it (zkr_key_replicate) to run two shares on one device", j); return ZKR_ERR_ARG; }
  }
  if (count == 0) return 0;
  const size_t used = n_keys < count ? n_keys : count;
  std::vector<Share> sh(used);
  for (size_t i = 0; i < count; i++) {
    Share &s = sh[i % used];
    s.wits.push_back(witnesses[i]);
    if (r32s) { s.rs.insert(s.rs.end(), r32s + 32 * i, r32s + 32 * i + 32); s.ss.insert(s.ss.end(), s32s + 32 * i, s32s + 32 * i + 32); }
  }
  auto work = [&](size_t j) {
    Share &s = sh[j];
    s.proofs.resize(256 * s.wits.size());
    try {
      s.rc = call(keys[j], s.wits.data(), s.wits.size(), r32s ? s.rs.data() : nullptr, r32s ? s.ss.data() : nullptr, s.proofs.data());
      if (s.rc) s.err = zkr_last_error();  // the message is the worker thread's: carry it to the caller's
    } catch (const std::exception &e) {
      s.rc = ZKR_ERR_HIP;
      s.err = std::string("share ") + std::to_string(j) + ": " + e.what();
    }
  };
  // one host thread per key (each entry point binds its thread to the key's device); the caller's thread takes share 0.
  // A thread that cannot be started (EAGAIN under a pids limit) must not take the host process down through the C ABI:
  // its share runs on the caller's thread afterwards.
  std::vector<std::thread> thr;
  std::vector<size_t> inline_shares;
  for (size_t j = 1; j < used; j++) {
    try { thr.emplace_back(work, j); } catch (const std::system_error &) { inline_shares.push_back(j); }
  }
  work(0);
  for (size_t j : inline_shares) work(j);
  for (auto &t : thr) t.join();
  for (Share &s : sh) { explicit_bzero(s.rs.data(), s.rs.size()); explicit_bzero(s.ss.data(), s.ss.size()); }
  for (size_t j = 0; j < used; j++)
    if (sh[j].rc) { set_error("key %zu (device %d): %s", j, keys[j]->device, sh[j].err.c_str()); return sh[j].rc; }
  for (size_t i = 0; i < count; i++) memcpy(proofs_out + 256 * i, &sh[i % used].proofs[256 * (i / used)], 256);
  return 0;
}

}  // namespace

extern "C" {

int zkr_prove_batch_multi(zkr_key *const *keys, size_t n_keys, const void *const *witnesses_std, size_t witness_len, size_t count, const uint8_t *r32s,
                          const uint8_t *s32s, uint8_t *proofs_out) {
  return run_shares(keys, n_keys, witnesses_std, count, r32s, s32s, proofs_out,
                    [&](zkr_key *k, const void *const *w, size_t n, const uint8_t *r, const uint8_t *s, uint8_t *out) { return zkr_prove_batch(k, w, witness_len, n, r, s, out); });
}

int zkr_prove_batch_multi_device(zkr_key *const *keys, size_t n_keys, const void *const *d_witnesses_std, size_t count, const uint8_t *r32s, const uint8_t *s32s,
                                 uint8_t *proofs_out) {
  return run_shares(keys, n_keys, d_witnesses_std, count, r32s, s32s, proofs_out,
                    [&](zkr_key *k, const void *const *w, size_t n, const uint8_t *r, const uint8_t *s, uint8_t *out) { return zkr_prove_batch_device(k, w, n, r, s, nullptr, out); });
}

int zkr_key_device(const zkr_key *key) { return key ? key->device : -1; }

// ---------------------------------------------------------------- ONE proof over several GPUs
int zkr_key_shard_info(const zkr_key *key, uint32_t out[6]) {
  if (!key || !out) { set_error("null argument"); return ZKR_ERR_ARG; }
  out[0] = key->h.shard_part; out[1] = key->h.shard_parts;
  out[2] = key->h.sc_lo[0]; out[3] = key->h.sc_n[0]; out[4] = key->h.sc_lo[1]; out[5] = key->h.sc_n[1];
  return 0;
}

int zkr_key_shard(const zkr_key *src, unsigned part, unsigned parts, int device, zkr_key **out) {
  if (!src || !out) { set_error("null argument"); return ZKR_ERR_ARG; }
  if (parts < 1 || parts > 64 || part >= parts) { set_error("shard %u of %u: parts must be 1..64 and part below it", part, parts); return ZKR_ERR_ARG; }
  if (src->h.shard_parts != 1) { set_error("the key is itself a shard (%u of %u): shard the whole key", src->h.shard_part, src->h.shard_parts); return ZKR_ERR_ARG; }
  if (zkr_device_count() <= device || device < 0) { set_error("no HIP device %d (found %d)", device, zkr_device_count()); return ZKR_ERR_NO_DEVICE; }
  const ArenaHeader &sh = src->h;
  ArenaHeader h = sh;
  h.shard_part = part; h.shard_parts = parts;
  const uint32_t vec_len[2] = {sh.n, sh.m};
  for (int v = 0; v < 2; v++) {
    h.sc_lo[v] = (uint32_t)((uint64_t)vec_len[v] * part / parts);
    h.sc_n[v] = (uint32_t)((uint64_t)vec_len[v] * (part + 1) / parts) - h.sc_lo[v];
    if (h.sc_n[v] == 0) { set_error("%u parts for a vector of %u scalars: a shard would be empty", parts, vec_len[v]); return ZKR_ERR_ARG; }
  }
  // the points of the range: rank maps are increasing over the kept scalars (key_build numbers the points in scalar order), so a
  // scalar range owns a contiguous point range [pt_lo, pt_lo + count) of every table
  ZKR_HIP_CHECK(hipSetDevice(src->device));
  std::vector<uint32_t> rank[N_TABLES];
  uint32_t pt_lo[N_TABLES];
  for (int t = 0; t < N_TABLES; t++) {
    const int v = t == T_H ? 1 : 0;
    rank[t].assign(h.sc_n[v], RANK_NONE_U32);
    pt_lo[t] = 0;
    if (!sh.npts[t] || !h.sc_n[v]) { h.npts[t] = 0; h.rank_identity[t] = 0; continue; }
    ZKR_HIP_CHECK(hipMemcpy(rank[t].data(), src->arena + sh.off_rank[t] + (size_t)h.sc_lo[v] * 4, (size_t)h.sc_n[v] * 4, hipMemcpyDeviceToHost));
    uint32_t first = RANK_NONE_U32, last = 0, cnt = 0;
    for (uint32_t r : rank[t])
      if (r != RANK_NONE_U32) { if (first == RANK_NONE_U32) first = r; if (cnt && r <= last) { set_error("rank map of table %d is not increasing", t); return ZKR_ERR_BAD_KEY; } last = r; cnt++; }
    if (cnt && last - first + 1 != cnt) { set_error("table %d: the scalar range does not own a contiguous point range", t); return ZKR_ERR_BAD_KEY; }
    pt_lo[t] = cnt ? first : 0;
    h.npts[t] = cnt;
    bool ident = cnt == h.sc_n[v];
    for (uint32_t i = 0; i < (uint32_t)rank[t].size(); i++) {
      if (rank[t][i] != RANK_NONE_U32) rank[t][i] -= pt_lo[t];
      ident = ident && rank[t][i] == i;
    }
    h.rank_identity[t] = ident ? 1 : 0;
  }
  // tables that share a digit sort must still agree point for point (they do: the same scalar range of the same supports)
  if (h.share_ac && h.npts[T_A] != h.npts[T_C]) h.share_ac = 0;
  if (h.share_b && h.npts[T_B1] != h.npts[T_B2]) h.share_b = 0;
  // Window size of the shard: from ITS scalar count, as a key load would choose it.  A shard's point range shrinks with the
  // number of parts, its bucket sets do not: 1/8 of a 2^20 key at the whole key's c = 20 sorts 2^17 scalars into 2^19 buckets per
  // table (3 entries each) and then walks four reduction chains over all of them -- the chains were 2/3 of such a shard's 4.1 ms.
  // At c = 17 it is an ordinary 2^17-point key (2^16 buckets).  The levels 2^(ck) P are then rebuilt from level 0 on the shard's
  // device (msm_precompute, as a key load does); tables whose window does not change are copied level by level.
  bool rebuild[N_TABLES];
  for (int t = 0; t < N_TABLES; t++) {
    const uint32_t c = (uint32_t)msm_plan(rank_entries(h, t), h.npts[t], 0).c;
    rebuild[t] = c != sh.win_c[t];
    h.win_c[t] = c;
  }
  arena_layout(h);
  // peer access between the shard's device and every other one, both ways, BEFORE anything of the shard is allocated: its
  // siblings' kernels read and write its vectors directly when a sharded proof splits calcH (zkr_prove.hip calc_h_split)
  for (int d = 0, nd = zkr_device_count(); d < nd; d++)
    if (d != device) { (void)peer_direct(device, d); (void)peer_direct(d, device); }
  ZKR_HIP_CHECK(hipSetDevice(device));
  DevBuf buf;
  if (int rc = buf.alloc(h.total_len)) return rc;
  unsigned char *arena = buf.as<unsigned char>();
  ZKR_HIP_CHECK(hipMemset(arena, 0, h.total_len));
  ZKR_HIP_CHECK(hipMemcpy(arena, &h, sizeof(h), hipMemcpyHostToDevice));
  auto cp = [&](uint64_t dst, uint64_t from, size_t bytes) { return bytes ? copy_between_devices(arena + dst, device, src->arena + from, src->device, bytes) : 0; };
  int rc = cp(h.off_tw, sh.off_tw, (size_t)sh.m * 32);
  if (!rc) rc = cp(h.off_twl, sh.off_twl, sh.off_rowptr[0] - sh.off_twl);  // the small twiddle table up to the next section (alignment slack included)
  for (int s = 0; s < 2 && !rc; s++) {
    const uint32_t nnz = s == 0 ? sh.nnzA : sh.nnzB;
    rc = cp(h.off_rowptr[s], sh.off_rowptr[s], ((size_t)sh.m + 1) * 4);
    if (!rc) rc = cp(h.off_col[s], sh.off_col[s], (size_t)nnz * 4);
    if (!rc) rc = cp(h.off_coef[s], sh.off_coef[s], (size_t)nnz * 32);
    if (!rc) rc = cp(h.off_wide[s], sh.off_wide[s], (size_t)sh.n_wide[s] * 4);
  }
  for (int t = 0; t < N_TABLES && !rc; t++) {
    const size_t pb = t == T_B2 ? 128 : 64, K = (255 + sh.win_c[t] - 1) / sh.win_c[t];
    if (rebuild[t]) {  // level 0 of the range back in the key's wire radix, then the shard's own levels
      rc = cp(h.off_pts[t], sh.off_pts[t] + (size_t)pt_lo[t] * pb, (size_t)h.npts[t] * pb);
      if (!rc) rc = radix_convert(device, t == T_B2, arena + h.off_pts[t], h.npts[t], false);
      if (!rc) rc = msm_precompute(device, t == T_B2, arena + h.off_pts[t], h.npts[t], msm_plan(rank_entries(h, t), h.npts[t], (int)h.win_c[t]));
    } else {
      for (size_t k = 0; k < K && !rc; k++)  // level k of the range: 2^(ck) P_i for the shard's points
        rc = cp(h.off_pts[t] + k * h.npts[t] * pb, sh.off_pts[t] + (k * sh.npts[t] + pt_lo[t]) * pb, (size_t)h.npts[t] * pb);
    }
    if (!rc && !rank[t].empty()) {
      ZKR_HIP_CHECK(hipSetDevice(device));
      ZKR_HIP_CHECK(hipMemcpy(arena + h.off_rank[t], rank[t].data(), rank[t].size() * 4, hipMemcpyHostToDevice));
    }
  }
  if (rc) return rc;
  zkr_key *k = nullptr;
  rc = zkr_key_adopt_arena(arena, h.total_len, device, &k);
  if (rc) return rc;
  k->owns_arena = true;
  buf.release();
  *out = k;
  return 0;
}

}  // extern "C"

namespace {
// phase times of the calling thread's last sharded proof with a split calcH (zkr_prove_sharded_split_stats)
thread_local double last_split_phase_ms[8][8];
thread_local unsigned last_split_parts = 0;
// which form the calling thread's last sharded proof took, and why (zkr_prove_sharded_last_form)
thread_local int last_form = ZKR_SHARDED_NONE;
thread_local char last_reason[256] = "";
void set_form(int form, const char *fmt, ...) {
  last_form = form;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_reason, sizeof(last_reason), fmt, ap);
  va_end(ap);
}

// every stream of every shard idle: after a failed group the stragglers' CROSS kernels (which write the OTHER shards' vectors) have
// ended before the caller -- who still holds the shards' turns -- lets anybody reuse a proof slot of these keys
void drain_shards(zkr_key *const *shards, size_t parts) {
  for (size_t i = 0; i < parts; i++) {
    zkr_key *k = shards[i];
    if (hipSetDevice(k->device) != hipSuccess) continue;
    (void)hipStreamSynchronize(k->stream);
    (void)hipStreamSynchronize(k->prep_stream);
    for (int j = 0; j < k->n_all; j++) (void)hipStreamSynchronize(k->red_stream[j]);
  }
  (void)hipGetLastError();
}

// One pass over the shards, one host thread each: split = calcH split over them (the caller has checked the preconditions and
// holds the shards' turns), else every shard computes h for itself.  *ran_split: whether the split really ran (false when a
// thread could not be started).
template <class Partial>
int run_shards_once(zkr_key *const *shards, size_t parts, bool split, int klog, std::vector<uint8_t> &partials, Partial partial, bool *ran_split, bool turns_held) {
  std::vector<int> rcs(parts, 0);
  std::vector<std::string> errs(parts);
  ShardGroup group;
  group.parts = (unsigned)parts;
  group.vecs.resize(parts);
  group.klog = klog;
  group.split_h = split;
  std::vector<char> threaded(parts, 0);
  auto work = [&](size_t i) {
    shard_group = group.split_h && threaded[i] ? &group : nullptr;  // a shard run inline after the others cannot meet them at a barrier
    shard_group_part = (unsigned)i;
    shard_turn_held = turns_held;  // the caller holds every shard's turn (a split pass, also when the split is then given up: no thread; the first-use check's second pass)
    try {
      rcs[i] = partial(i, &partials[i * ZKR_PARTIAL_BYTES]);
      if (rcs[i]) errs[i] = zkr_last_error();
    } catch (const std::exception &e) { rcs[i] = ZKR_ERR_HIP; errs[i] = e.what(); }
    if (rcs[i]) group.abort();
    shard_group = nullptr;
    shard_turn_held = false;
  };
  // a split calcH needs every shard on a thread of its own, all at once: when a thread cannot be started the group falls back to
  // replicated calcH BEFORE anybody runs (the threads wait for the decision)
  std::vector<std::thread> thr;
  std::vector<size_t> inline_parts;
  std::mutex go_mu;
  std::condition_variable go_cv;
  bool go = false;
  auto gated = [&](size_t i) {
    { std::unique_lock<std::mutex> lk(go_mu); go_cv.wait(lk, [&] { return go; }); }
    work(i);
  };
  threaded[0] = 1;
  for (size_t i = 1; i < parts; i++) {
    try { thr.emplace_back(gated, i); threaded[i] = 1; } catch (const std::system_error &) { inline_parts.push_back(i); }
  }
  if (!inline_parts.empty()) group.split_h = false;
  { std::lock_guard<std::mutex> lk(go_mu); go = true; }
  go_cv.notify_all();
  work(0);
  for (size_t i : inline_parts) work(i);
  for (auto &t : thr) t.join();
  *ran_split = group.split_h;
  if (group.split_h) memcpy(last_split_phase_ms, group.phase_ms, sizeof(last_split_phase_ms));
  last_split_parts = group.split_h ? (unsigned)parts : 0;
  bool any = false;
  for (size_t i = 0; i < parts; i++) any = any || rcs[i];
  if (any && split) drain_shards(shards, parts);  // the failing shard drained its own streams (prove_submit_group); its siblings' cross passes may still run
  // the shard that failed first, not one that gave up because of it
  for (int pass = 0; pass < 2; pass++)
    for (size_t i = 0; i < parts; i++)
      if (rcs[i] && (pass == 1 || errs[i].find("another shard of the proof failed") == std::string::npos)) {
        set_error("shard %zu (device %d): %s", i, shards[i]->device, errs[i].c_str());
        return rcs[i];
      }
  return 0;
}

template <class Partial>
int run_sharded(zkr_key *const *shards, size_t parts, const uint8_t *r32, const uint8_t *s32, uint8_t *proof_out, Partial partial) {
  set_form(ZKR_SHARDED_NONE, "");
  last_split_parts = 0;
  if (!shards || parts == 0 || !proof_out) { set_error("null argument"); return ZKR_ERR_ARG; }
  for (size_t i = 0; i < parts; i++) {
    if (!shards[i]) { set_error("shard %zu is null", i); return ZKR_ERR_ARG; }
    const ArenaHeader &h = shards[i]->h;
    if (h.shard_parts != parts || h.shard_part != i) { set_error("shards[%zu] is part %u of %u: pass part i of %zu at position i", i, h.shard_part, h.shard_parts, parts); return ZKR_ERR_ARG; }
    if (h.n != shards[0]->h.n || h.m != shards[0]->h.m || memcmp(h.delta1, shards[0]->h.delta1, 64)) { set_error("shard %zu belongs to another key", i); return ZKR_ERR_ARG; }
  }
  std::vector<uint8_t> partials(parts * ZKR_PARTIAL_BYTES);
  // The shards run concurrently, so calcH can be split over them instead of repeated by every one (zkr_prove.hip calc_h_split):
  // a power of two of shards, each owning one aligned block of h with enough columns for its share of the cross passes, and
  // every device able to read and write every other's memory.  ZKR_SHARD_SPLIT_H=0: every shard computes h for itself; =1: split
  // whenever the preconditions hold, unchecked; unset: split, but shards that sit on DIFFERENT devices -- where the cross passes
  // really cross a link, ordered only by stream synchronisation + a host barrier -- prove their first proof BOTH ways and keep the
  // split only if the two forms' sums agree (ZKR_SHARD_SPLIT_CHECK=1 forces that check for shards on one device: tests).
  const char *e = getenv("ZKR_SHARD_SPLIT_H");
  const int policy = e ? (atoi(e) == 0 ? 0 : 1) : 2;  // 0 never, 1 always, 2 checked
  const uint32_t m = shards[0]->h.m;
  int klog = 0;
  while ((1u << klog) < parts) klog++;
  char why[200] = "";
  bool ok = true;
  if (policy == 0) { ok = false; snprintf(why, sizeof(why), "ZKR_SHARD_SPLIT_H=0"); }
  else if (parts < 2 || parts > 8 || (1u << klog) != parts) { ok = false; snprintf(why, sizeof(why), "%zu shards: the split needs 2, 4 or 8", parts); }
  else if ((m >> (2 * klog)) < 64) { ok = false; snprintf(why, sizeof(why), "domain 2^%u too small for %zu shards (fewer than 64 columns per cross pass)", shards[0]->h.logm, parts); }
  for (size_t i = 0; i < parts && ok; i++)
    if (shards[i]->h.sc_n[1] != m >> klog || shards[i]->h.sc_lo[1] != (uint32_t)i * (m >> klog)) { ok = false; snprintf(why, sizeof(why), "shard %zu does not own block %zu of h", i, i); }
  bool distinct = false;
  for (size_t i = 1; i < parts; i++) distinct = distinct || shards[i]->device != shards[0]->device;
  for (size_t i = 0; i < parts && ok; i++)
    for (size_t j = 0; j < parts && ok; j++)
      if (!peer_direct(shards[i]->device, shards[j]->device)) { ok = false; snprintf(why, sizeof(why), "no peer access from device %d to device %d", shards[i]->device, shards[j]->device); }
  const char *ce = getenv("ZKR_SHARD_SPLIT_CHECK");
  const int force_check = ce ? atoi(ce) : 0;  // 1: check although the shards share a device; 2: and pretend the forms disagreed (tests of the fallback)
  // A split calcH makes the shards' threads wait for one another INSIDE their enqueue, each holding its shard's lock: two such
  // proofs on the same shards at once could wait for each other's locks for ever.  They take turns (locks in address order,
  // exclusive; a shard proving on its own -- zkr_prove_partial outside a split group -- takes its own shard's turn shared).
  std::vector<std::unique_lock<std::shared_mutex>> turn;
  if (ok) {
    std::vector<zkr_key *> order(shards, shards + parts);
    std::sort(order.begin(), order.end());
    for (zkr_key *k : order) turn.emplace_back(k->split_mu);
  }
  // The state of the first-use check belongs to the shard SET: read under the turns (a concurrent first caller has finished its
  // check by now) and over every shard -- a set that reuses a checked shard with new siblings is unchecked, one member that saw
  // the forms disagree condemns the set.
  int state = 1;
  for (size_t i = 0; i < parts; i++) {
    const int st = shards[i]->split_checked.load();
    if (st == 2) { state = 2; break; }
    if (st == 0) state = 0;
  }
  if (ok && policy == 2 && state == 2) { ok = false; turn.clear(); snprintf(why, sizeof(why), "the split form failed its first-use check on these shards (it disagreed with the replicated one)"); }
  const bool check = ok && policy == 2 && state == 0 && (distinct || force_check);
  bool ran_split = false;
  int rc;
  if (check) {
    // first use on these devices: the split form, then the replicated one (the turns stay held: nobody else starts on these shards
    // in between); their sums must be the same group elements (compared through the proof both assemble with a fixed blinding --
    // the XYZZ coordinates of a sum depend on the order of additions)
    std::vector<uint8_t> split_partials(parts * ZKR_PARTIAL_BYTES);
    uint8_t one[32] = {1}, pa[256], pb[256];
    rc = run_shards_once(shards, parts, true, klog, split_partials, partial, &ran_split, true);
    double keep_ms[8][8];
    memcpy(keep_ms, last_split_phase_ms, sizeof(keep_ms));
    const bool split_ran = rc == 0 && ran_split;
    bool dummy = false;
    int rc2 = run_shards_once(shards, parts, false, klog, partials, partial, &dummy, true);
    if (rc2) return rc2;
    // compared: the split pass ran and both sets of sums assemble; a pass that could not run (a transient HIP failure, no thread)
    // or sums that do not assemble with r = s = 1 decide nothing -- the set stays unchecked and the next proof tries again
    bool compared = false, same = false;
    if (split_ran && zkr_prove_combine(shards[0], split_partials.data(), parts, one, one, pa) == 0 && zkr_prove_combine(shards[0], partials.data(), parts, one, one, pb) == 0) {
      compared = true;
      same = memcmp(pa, pb, 256) == 0 && force_check != 2;
    }
    if (compared) {
      for (size_t i = 0; i < parts; i++) shards[i]->split_checked.store(same ? 1 : 2);
      if (same) {
        memcpy(last_split_phase_ms, keep_ms, sizeof(keep_ms));
        last_split_parts = (unsigned)parts;
        set_form(ZKR_SHARDED_SPLIT_H, "split calcH: first use on devices %d..%d proved both ways, sums identical; proofs from now on split", shards[0]->device, shards[parts - 1]->device);
      } else {
        fprintf(stderr, "zkr: sharded proof: the split calcH DISAGREED with the replicated form on first use over devices %d..%d -- every shard computes h for itself from now on (ZKR_SHARD_SPLIT_H=1 forces the split)\n",
                shards[0]->device, shards[parts - 1]->device);
        set_form(ZKR_SHARDED_REPLICATED_H, "replicated calcH: the split form disagreed with the replicated one on first use over devices %d..%d", shards[0]->device, shards[parts - 1]->device);
      }
    } else {
      set_form(ZKR_SHARDED_REPLICATED_H, "replicated calcH: the split form's first-use check could not be made (%s); it is tried again with the next proof",
               rc ? "the split pass failed" : !ran_split ? "a shard's host thread could not be started" : "the sums did not assemble");
    }
    turn.clear();
    return zkr_prove_combine(shards[0], partials.data(), parts, r32, s32, proof_out);  // the replicated pass' sums either way
  }
  rc = run_shards_once(shards, parts, ok, klog, partials, partial, &ran_split, ok);
  if (rc) return rc;
  if (ran_split) set_form(ZKR_SHARDED_SPLIT_H, "split calcH: %s", policy == 1 ? "ZKR_SHARD_SPLIT_H=1" : distinct ? "checked against the replicated form on first use" : "all shards on one device");
  else set_form(ZKR_SHARDED_REPLICATED_H, "replicated calcH: %s", ok ? "a shard's host thread could not be started" : why);
  return zkr_prove_combine(shards[0], partials.data(), parts, r32, s32, proof_out);
}
}  // namespace

extern "C" {

int zkr_prove_sharded(zkr_key *const *shards, size_t parts, const void *witness_std, size_t witness_len, const uint8_t *r32, const uint8_t *s32,
                      uint8_t proof_out[256]) {
  if (!witness_std) { set_error("null argument"); return ZKR_ERR_ARG; }
  return run_sharded(shards, parts, r32, s32, proof_out, [&](size_t i, uint8_t *out) { return zkr_prove_partial(shards[i], witness_std, witness_len, out); });
}
int zkr_prove_sharded_split_stats(unsigned *parts_out, double phase_ms_out[64]) {
  if (!parts_out || !phase_ms_out) { set_error("null argument"); return ZKR_ERR_ARG; }
  *parts_out = last_split_parts;
  memcpy(phase_ms_out, last_split_phase_ms, sizeof(last_split_phase_ms));
  return 0;
}
int zkr_prove_sharded_last_form(int *form_out, char *reason_out, size_t reason_len) {
  if (!form_out) { set_error("null argument"); return ZKR_ERR_ARG; }
  *form_out = last_form;
  if (reason_out && reason_len) { strncpy(reason_out, last_reason, reason_len - 1); reason_out[reason_len - 1] = 0; }
  return 0;
}
int zkr_key_replication(const zkr_key *key, int *mode_out, int *peer_direct_out) {
  if (!key || !mode_out) { set_error("null argument"); return ZKR_ERR_ARG; }
  *mode_out = key->replica_mode;
  if (peer_direct_out) *peer_direct_out = key->replica_direct ? 1 : 0;
  return 0;
}
int zkr_prove_sharded_device(zkr_key *const *shards, size_t parts, const void *const *d_witnesses_std, const uint8_t *r32, const uint8_t *s32,
                             uint8_t proof_out[256]) {
  if (!d_witnesses_std) { set_error("null argument"); return ZKR_ERR_ARG; }
  return run_sharded(shards, parts, r32, s32, proof_out, [&](size_t i, uint8_t *out) { return zkr_prove_partial_device(shards[i], d_witnesses_std[i], nullptr, out); });
}

}  // extern "C"
