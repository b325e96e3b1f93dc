// zkr_multi.hip -- several GPUs of one node behind the C ABI, without torch.distributed: key replication device to
// device and a batch call that shards independent proofs over the replicas, one host thread per key.
//
// SURVEY.md 8(b) "Threading" (batch-of-64 driver: one host thread per GPU, hipSetDevice per worker) and 8(e): the rollup
// operator's host is Node (/root/reference/operator/src/snarks/common.ts:23-29, scripts/index.js:40-46) -- one process that
// awaits proofs -- so the proof-level sharding of BASELINE configs[3] has to be reachable from ONE process through the
// library the N-API shim binds, not only from one-process-per-GPU launchers (python/zkr_hip/batch.py keeps that form for
// torchrun + RCCL).
#include <string.h>
#include <exception>
#include <string>
#include <thread>
#include <vector>
#include "zkr_internal.hpp"

using namespace zkr;

namespace {

constexpr size_t PEER_PIECE = (size_t)1 << 30;  // bytes per peer copy (arenas reach 78 GB at 2^24)

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

// direct loads/stores between the two devices (xGMI on an MI355X node).  Without it hipMemcpyPeer still works -- staged
// through host memory by the runtime -- which is what the compact form is for.
bool peer_direct(int dst_dev, int src_dev) {
  if (dst_dev == src_dev) return true;
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
    buf.release();
    *out = k;
    return 0;
  }
  // compact form: base points + QAP rows (1/10 of the bytes), window levels and twiddles rebuilt on the receiver
  void *base = nullptr;
  size_t base_len = 0;
  if (int rc = zkr_key_base_arena(const_cast<zkr_key *>(src), &base, &base_len)) return rc;
  if (dst_device == src->device) return zkr_key_adopt_base_arena(base, base_len, dst_device, out);
  ZKR_HIP_CHECK(hipSetDevice(dst_device));
  DevBuf tmp;
  if (int rc = tmp.alloc(base_len)) return rc;
  if (int rc = copy_between_devices(tmp.p, dst_device, base, src->device, base_len)) return rc;
  return zkr_key_adopt_base_arena(tmp.p, base_len, dst_device, out);
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

}  // extern "C"
