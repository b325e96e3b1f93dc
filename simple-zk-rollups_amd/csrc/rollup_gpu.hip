// rollup_gpu.hip -- MiMCSponge-220 on the GPU, batch-parallel: the hash of the reference's balance tree and leaves
// (/root/reference/operator/src/utils/crypto.ts:28-38 `multiHash` / `hashLeftRight`, prover/circuits/hasher.circom:3-30;
// operator/src/utils/merkletree.ts:44-83 builds the tree with it).  SURVEY 8(f-3): "batch-parallel ... on GPU" -- one thread
// per hash, 220 rounds of x -> x^5 in Fr Montgomery arithmetic (3 multiplications per round), so a tree of 2^20 accounts
// (2^21 - 1 hashes of two elements) is 2.8 * 10^9 multiplications: tens of milliseconds here, minutes in the circomlib
// BigInt code.  Host counterpart and round constants: rollup.cpp; parity against oracle/rollup.py in tests/test_gpu_rollup.py.
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include "zkr_internal.hpp"
#include "rollup_witness.hpp"

namespace zkr {
const Fr *mimc_round_constants();  // rollup.cpp
void rollup_device_constants(Fr *a, Fr *d, uint32_t suborder_m1[8], Fr *b8x253, Fr *b8y253);
uint32_t rollup_tx_private_count(uint32_t depth);
uint32_t rollup_n_public(uint32_t batch, uint32_t depth);
int rollup_check_geometry(uint32_t batch, uint32_t depth);

__constant__ uint32_t c_mimc[MIMC_ROUNDS * 8];

__device__ __forceinline__ Fr mimc_const(int i) {
  Fr k;
#pragma unroll
  for (int j = 0; j < 8; j++) k.v[j] = c_mimc[8 * i + j];
  return k;
}
__device__ __forceinline__ void mimc_feistel(Fr &xl, Fr &xr) {  // key 0 (hasher.circom:21)
  for (int i = 0; i < MIMC_ROUNDS; i++) {
    Fr t = add(xl, mimc_const(i));
    Fr t2 = sqr(t);
    Fr t5 = mul(sqr(t2), t);
    if (i < MIMC_ROUNDS - 1) {
      Fr nl = add(xr, t5);
      xr = xl;
      xl = nl;
    } else {
      xr = add(xr, t5);
    }
  }
}
__device__ __forceinline__ Fr load_std_as_mont(const Fr *p) {
  Fr a = *p;
  // operands of any size are taken mod r as the reference's field operations do: 2^256 < 6 r
  for (int k = 0; k < 5; k++) {
    bool ge = true;
    for (int i = 7; i >= 0; i--)
      if (a.v[i] != FrParams::P[i]) { ge = a.v[i] > FrParams::P[i]; break; }
    if (!ge) break;
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) a.v[i] = __builtin_subc(a.v[i], FrParams::P[i], br, &br);
  }
  return to_mont(a);
}

// out[t] = multiHash(in[t * arity .. t * arity + arity))   (standard form in and out)
static __global__ __launch_bounds__(256) void mimc_multihash_kernel(const Fr *in, size_t count, uint32_t arity, Fr *out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  Fr r = Fr::zero(), c = Fr::zero();
  for (uint32_t i = 0; i < arity; i++) {
    r = add(r, load_std_as_mont(in + t * arity + i));
    mimc_feistel(r, c);
  }
  out[t] = from_mont(r);
}

static int upload_constants(int device) {
  static std::mutex mu;
  static bool done[64] = {false};
  std::lock_guard<std::mutex> lk(mu);
  if (device < 64 && done[device]) return 0;
  ZKR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_mimc), mimc_round_constants(), MIMC_ROUNDS * 32));
  if (device < 64) done[device] = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------ witnesses on the GPU
// SURVEY 8(f-3) "batch-parallel across txs on GPU": the witness of BatchProcessTx(batch, depth) for MANY rollup batches at
// once, one thread per transaction, left in HBM for zkr_prove_batch_device.  The program below is the value side of the
// gadget program of rollup.cpp (processtx.circom:10-193 over eddsa.circom:12-139, merkletree.circom:5-84,
// hasher.circom:3-30): the same statements in the same order, every allocated signal written where rollup.cpp's Builder
// puts it, so the bytes equal zkr_rollup_witness's (tests/test_gpu_rollup.py).  A transaction is ~6 * 10^4 dependent field
// multiplications (74 MiMC permutations, two 254-step scalar multiplications): one lane takes ~35 ms for it, and the
// transactions of all batches run side by side -- 512 of them (256 batches of tx.circom) in the time of one, where the
// host builder needs ~13 ms of one core per batch.  Values are Montgomery while they are computed; a second kernel lays the
// vector out as binarifyWitness does (32 B standard form per signal, operator/src/utils/binarify.ts:10-48).
// One thread per transaction.  inputs: per batch the n_public - 1 input signals (standard form, circom's order: rollup.cpp
// Layout); wit: the witnesses, nv signals per batch -- a transaction writes its K private signals (Montgomery for now) into its
// slice behind the public part; roots / errs: per transaction.
static __global__ __launch_bounds__(64) void rollup_witness_tx_kernel(const Fr *inputs, uint32_t n_batches, uint32_t batch, uint32_t depth, uint32_t K, TxConsts k,
                                                                      Fr *wit, Fr *workspace, Fr *roots, uint32_t *errs) {
  __builtin_amdgcn_s_setprio(3);  // a few wavefronts with a dependent chain of ~15 ms each: they win the issue slot against the bulk kernels of proofs in flight
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, part = blockIdx.y;  // a wavefront = one part of 64 transactions
  const uint32_t n_tx = n_batches * batch;
  if (t >= n_tx) return;
  const uint32_t bi = t / batch, i = t % batch;
  const uint32_t p = tx_layout(batch, depth).p;
  Fr root;
  errs[(size_t)part * n_tx + t] = tx_witness(inputs + (size_t)bi * (p - 1), batch, depth, i, K, k, wit + (size_t)bi * ((size_t)p + 1 + (size_t)batch * K) + p + 1 + (size_t)i * K,
                                             workspace + (size_t)t * TX_WS_ELEMS, &root, (int)part);
  if (part == TX_PARTS - 1) roots[t] = root;
}
// The comparison of the signature equation's two sides, which parts 0 and 1 computed apart (rollup_witness.hpp tx_finish): a thread
// per transaction, a handful of field operations when the signature holds.
static __global__ __launch_bounds__(64) void rollup_witness_finish_kernel(uint32_t n_batches, uint32_t batch, uint32_t depth, uint32_t K, Fr *wit, uint32_t *errs) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, n_tx = n_batches * batch;
  if (t >= n_tx) return;
  const uint32_t bi = t / batch, i = t % batch;
  const uint32_t p = tx_layout(batch, depth).p;
  errs[(size_t)TX_PARTS * n_tx + t] = tx_finish(wit + (size_t)bi * ((size_t)p + 1 + (size_t)batch * K) + p + 1 + (size_t)i * K, depth, K);
}
// The vector as binarifyWitness lays it out: signal 0 = 1, signal 1 = the last transaction's root, the inputs as given, the
// private signals converted to standard form in place.  Thread per signal of every batch; the first thread of a batch also
// checks the root chain (batchprocesstx.circom:67-69).
static __global__ __launch_bounds__(256) void rollup_witness_layout_kernel(const Fr *inputs, uint32_t n_batches, uint32_t batch, uint32_t depth, uint32_t K,
                                                                          const Fr *roots, Fr *wit, uint32_t *chain_errs) {
  const uint32_t p = 1 + batch * (8 + 2 + 1 + 1 + 2 + 1 + 1 + 1 + 1 + 3 * depth);
  const size_t nv = (size_t)p + 1 + (size_t)batch * K;
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nv * n_batches) return;
  const size_t bi = g / nv, s = g % nv;
  Fr v;
  if (s == 0) {
    v = Fr::zero();
    v.v[0] = 1;
    uint32_t bad = 0;  // transaction i starts from the root transaction i - 1 produced (the inputs' root array starts at signal 2)
    for (uint32_t i = 1; i < batch && !bad; i++) {
      Fr want;
      tx_read_input(inputs + bi * (p - 1) + i, want);
      if (!(roots[bi * batch + i - 1] == want)) bad = i;
    }
    chain_errs[bi] = bad;
  } else if (s == 1) {
    v = from_mont(roots[bi * batch + batch - 1]);
  } else if (s <= p) {
    v = inputs[bi * (p - 1) + (s - 2)];
  } else {
    v = from_mont(wit[g]);
  }
  wit[g] = v;
}
struct DevView {  // a typed look at device memory owned elsewhere
  void *p;
  template <class T> T *as() const { return static_cast<T *>(p); }
};
}  // namespace zkr

// Host fast path of zkr_rollup_witness (rollup.cpp batch_witness): the value program on host threads, one task per
// (transaction, part) as the GPU builder's wavefronts run it, then the signals to standard form in slices.  inputs_std: the p - 1
// circuit inputs (32 B standard form, already checked < r); out: (p + 1 + batch * K) x 32 B.  Returns true when the witness is
// complete and every statement holds; false (nothing promised about `out`) when any statement fails -- the caller then runs the
// gadget builder, which names the violated statement the way it always did.
namespace zkr {
bool rollup_witness_fast_host(uint32_t batch, uint32_t depth, const uint8_t *inputs_std, uint8_t *out_bytes) {
  const uint32_t p = rollup_n_public(batch, depth), K = rollup_tx_private_count(depth);
  std::vector<Fr> tab(2 * 253), in(p - 1), w((size_t)batch * K), roots(batch);
  std::vector<uint32_t> errs((size_t)batch * (TX_PARTS + 1), 0);
  std::vector<std::atomic<uint32_t>> parts_done(batch);
  for (auto &d : parts_done) d.store(0);
  TxConsts k;
  rollup_device_constants(&k.a, &k.d, k.suborder_m1, tab.data(), tab.data() + 253);
  k.b8x = tab.data(), k.b8y = tab.data() + 253, k.mimc = mimc_round_constants();
  memcpy(in.data(), inputs_std, (size_t)(p - 1) * 32);
  Fr *out = reinterpret_cast<Fr *>(out_bytes);
  const uint32_t tasks = batch * TX_PARTS;
  // Threads of this call: at most twelve (tx.circom is twelve tasks), shared out between the calls that are inside this function
  // at the same time -- a caller that already builds many witnesses side by side (the host-witness pipeline: fourteen callers on
  // a sixteen-CPU quota) gets one thread per call instead of 168 that wait for each other
  static std::atomic<int> calls_inside{0};
  struct Inside { Inside() { calls_inside.fetch_add(1); } ~Inside() { calls_inside.fetch_sub(1); } } inside;
  unsigned hw = std::thread::hardware_concurrency();
  unsigned cap = 12;
  if (const char *e = getenv("ZKR_WITNESS_THREADS")) { int v = atoi(e); if (v >= 1) hw = cap = (unsigned)v; }
  const unsigned share = std::max(1u, std::min(hw ? hw : 1u, cap) / (unsigned)std::max(1, calls_inside.load()));
  const uint32_t nthreads = std::max<uint32_t>(1, std::min<uint32_t>(tasks, share));
  const size_t total = (size_t)batch * K, slice = 4096, nslices = (total + slice - 1) / slice;
  std::atomic<uint32_t> next_task{0}, next_slice{0}, arrived{0};
  std::mutex bar_mu;
  std::condition_variable bar_cv;
  auto worker = [&] {
    {
      std::vector<Fr> ws(TX_WS_ELEMS);
      for (uint32_t t; (t = next_task.fetch_add(1)) < tasks;) {
        const uint32_t i = t / TX_PARTS, part = t % TX_PARTS;
        Fr r;
        errs[t] = tx_witness(in.data(), batch, depth, i, K, k, w.data() + (size_t)i * K, ws.data(), &r, (int)part);
        if (part == TX_PARTS - 1) roots[i] = r;
        if (parts_done[i].fetch_add(1) + 1 == TX_PARTS)  // the transaction's last part: both sides of its signature equation are in place
          errs[(size_t)batch * TX_PARTS + i] = tx_finish(w.data() + (size_t)i * K, depth, K);
      }
    }
    {  // every signal is in place before any slice is converted (a blocking barrier: waiting threads do not spin)
      std::unique_lock<std::mutex> lk(bar_mu);
      if (arrived.fetch_add(1) + 1 == nthreads) bar_cv.notify_all();
      else bar_cv.wait(lk, [&] { return arrived.load() >= nthreads; });
    }
    for (uint32_t c; (c = next_slice.fetch_add(1)) < nslices;) {
      const size_t lo = (size_t)c * slice, hi = std::min(total, lo + slice);
      for (size_t s = lo; s < hi; s++) out[(size_t)p + 1 + s] = from_mont(w[s]);
    }
  };
  if (nthreads <= 1) {
    worker();
  } else {
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < nthreads; t++) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
  }
  for (uint32_t e : errs)
    if (e) return false;
  for (uint32_t i = 1; i < batch; i++) {  // the root chain (batchprocesstx.circom:67-69)
    Fr want;
    if (!tx_read_input(&in[i], want) || !(roots[i - 1] == want)) return false;
  }
  Fr one = Fr::zero();
  one.v[0] = 1;
  out[0] = one;
  out[1] = from_mont(roots[batch - 1]);
  for (uint32_t s = 2; s <= p; s++) out[s] = in[s - 2];
  return true;
}
}  // namespace zkr

using namespace zkr;

extern "C" {

int zkr_mimcsponge_multihash_batch(const void *inputs_std, size_t count, unsigned arity, void *out_std, int device) {
  if (!inputs_std || !out_std || arity < 1 || arity > 64) { set_error("bad argument"); return ZKR_ERR_ARG; }
  if (zkr_device_count() <= device || device < 0) { set_error("no HIP device %d; the batch hash runs on the GPU (zkr_mimcsponge_multihash is the host call)", device); return ZKR_ERR_NO_DEVICE; }
  if (count == 0) return ZKR_OK;
  ZKR_HIP_CHECK(hipSetDevice(device));
  int rc = upload_constants(device);
  if (rc) return rc;
  Fr *d_in = nullptr, *d_out = nullptr;
  ZKR_HIP_CHECK(hipMalloc(&d_in, count * arity * 32));
  if (hipMalloc(&d_out, count * 32) != hipSuccess) { hipFree(d_in); set_error("out of device memory"); return ZKR_ERR_HIP; }
  hipError_t e = hipMemcpy(d_in, inputs_std, count * arity * 32, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    mimc_multihash_kernel<<<(unsigned)((count + 255) / 256), 256>>>(d_in, count, arity, d_out);
    e = hipMemcpy(out_std, d_out, count * 32, hipMemcpyDeviceToHost);
  }
  hipFree(d_in);
  hipFree(d_out);
  if (e != hipSuccess) { set_error("batch hash failed: %s", hipGetErrorString(e)); return ZKR_ERR_HIP; }
  return ZKR_OK;
}

int zkr_rollup_witness_batch_device(uint32_t batch, uint32_t depth, const uint8_t *inputs, size_t n_inputs, size_t n_batches, void *d_witnesses, int device) {
  if ((!inputs && n_batches) || !d_witnesses) { set_error("null argument"); return ZKR_ERR_ARG; }
  int rc = rollup_check_geometry(batch, depth);
  if (rc) return rc;
  if (zkr_device_count() <= device || device < 0) { set_error("no HIP device %d; the batch witness builder runs on the GPU (zkr_rollup_witness is the host call)", device); return ZKR_ERR_NO_DEVICE; }
  const uint32_t p = rollup_n_public(batch, depth), K = rollup_tx_private_count(depth);
  if (n_inputs != p - 1) { set_error("rollup circuit (%u, %u) takes %u inputs, got %zu", batch, depth, p - 1, n_inputs); return ZKR_ERR_ARG; }
  if (n_batches == 0) return ZKR_OK;
  if (n_batches * (size_t)batch > (1u << 22)) { set_error("too many transactions in one call (%zu batches of %u)", n_batches, batch); return ZKR_ERR_ARG; }
  ZKR_HIP_CHECK(hipSetDevice(device));
  if ((rc = upload_constants(device))) return rc;
  // curve constants, the fixed-base table and the round constants, once per device
  static std::mutex mu;
  static Fr *d_tab[64] = {nullptr};
  TxConsts k;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (device >= 64) { set_error("device index out of range"); return ZKR_ERR_ARG; }
    std::vector<Fr> tab(2 * 253 + MIMC_ROUNDS);
    rollup_device_constants(&k.a, &k.d, k.suborder_m1, tab.data(), tab.data() + 253);
    memcpy(tab.data() + 506, mimc_round_constants(), MIMC_ROUNDS * 32);
    if (!d_tab[device]) {
      Fr *d = nullptr;
      ZKR_HIP_CHECK(hipMalloc(&d, tab.size() * 32));
      if (hipMemcpy(d, tab.data(), tab.size() * 32, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); set_error("upload of the witness builder's tables failed"); return ZKR_ERR_HIP; }
      d_tab[device] = d;
    }
  }
  k.b8x = d_tab[device], k.b8y = d_tab[device] + 253, k.mimc = d_tab[device] + 506;
  const size_t n_tx = n_batches * batch, nv = (size_t)p + 1 + (size_t)batch * K;
  // Scratch kept per device and only ever grown: hipMalloc / hipFree synchronise the whole device, i.e. every call would wait
  // for the proofs in flight on the key's streams (measured: the builder's 30 ms became the 63 ms of the prove call beside it).
  // One call per device at a time (the lock is held to the end of the call).
  struct Scratch { void *p = nullptr; size_t cap = 0; };
  static Scratch scratch[64][4];
  static std::mutex scratch_mu[64];
  std::lock_guard<std::mutex> scratch_lock(scratch_mu[device]);
  const uint32_t ROWS = TX_PARTS + 1;  // a row of statement codes per part, one for tx_finish
  const size_t want[4] = {n_batches * (size_t)(p - 1) * 32, n_tx * (size_t)TX_WS_ELEMS * 32, n_tx * 32, (ROWS * n_tx + n_batches) * 4};
  for (int j = 0; j < 4; j++) {
    Scratch &sc = scratch[device][j];
    if (sc.cap >= want[j]) continue;
    if (sc.p) hipFree(sc.p);
    sc.p = nullptr, sc.cap = 0;
    ZKR_HIP_CHECK(hipMalloc(&sc.p, want[j] + want[j] / 2));
    sc.cap = want[j] + want[j] / 2;
  }
  const DevView b_in{scratch[device][0].p}, b_ws{scratch[device][1].p}, b_roots{scratch[device][2].p}, b_errs{scratch[device][3].p};
  // Its own stream (non-blocking, highest priority): the builder is a handful of long-running wavefronts that must neither wait
  // for nor hold up the proofs in flight on the key's streams; nothing here synchronises the device.
  // Made once per device and kept (the scratch lock above serialises the calls): streams that come and go reshuffle the
  // runtime's hardware queues under the provers' streams (zkr_key.hip DeviceStreams).
  struct StreamRef { hipStream_t s = nullptr; };
  static StreamRef kept[64];
  StreamRef &sg = kept[device];
  if (!sg.s) {
    int prio_lo = 0, prio_hi = 0;
    ZKR_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    ZKR_HIP_CHECK(hipStreamCreateWithPriority(&sg.s, hipStreamNonBlocking, prio_hi));
  }
  ZKR_HIP_CHECK(hipMemcpyAsync(b_in.p, inputs, n_batches * (size_t)(p - 1) * 32, hipMemcpyHostToDevice, sg.s));
  uint32_t *d_errs = b_errs.as<uint32_t>(), *d_chain = d_errs + ROWS * n_tx;
  rollup_witness_tx_kernel<<<dim3((unsigned)((n_tx + 63) / 64), TX_PARTS), 64, 0, sg.s>>>(b_in.as<Fr>(), (uint32_t)n_batches, batch, depth, K, k, (Fr *)d_witnesses, b_ws.as<Fr>(), b_roots.as<Fr>(), d_errs);
  ZKR_HIP_CHECK(hipGetLastError());
  rollup_witness_finish_kernel<<<(unsigned)((n_tx + 63) / 64), 64, 0, sg.s>>>((uint32_t)n_batches, batch, depth, K, (Fr *)d_witnesses, d_errs);
  ZKR_HIP_CHECK(hipGetLastError());
  const size_t total = nv * n_batches;
  rollup_witness_layout_kernel<<<(unsigned)((total + 255) / 256), 256, 0, sg.s>>>(b_in.as<Fr>(), (uint32_t)n_batches, batch, depth, K, b_roots.as<Fr>(), (Fr *)d_witnesses, d_chain);
  ZKR_HIP_CHECK(hipGetLastError());
  std::vector<uint32_t> errs(ROWS * n_tx + n_batches);
  ZKR_HIP_CHECK(hipMemcpyAsync(errs.data(), d_errs, errs.size() * 4, hipMemcpyDeviceToHost, sg.s));
  ZKR_HIP_CHECK(hipStreamSynchronize(sg.s));
  for (size_t bi = 0; bi < n_batches; bi++) {  // first violated statement in circuit order, as zkr_rollup_witness reports it
    for (uint32_t i = 0; i < batch; i++) {
      const uint32_t chain = errs[ROWS * n_tx + bi];
      uint32_t e = 0;  // the codes are in program order: the smallest one over the parts is the transaction's first violation
      for (uint32_t part = 0; part < ROWS; part++) {
        const uint32_t ep = errs[(size_t)part * n_tx + bi * batch + i];
        if (ep && (!e || ep < e)) e = ep;
      }
      if (e == ST_INPUT_RANGE) { set_error("batch %zu: an input of transaction %u is not below r", bi, i); return ZKR_ERR_ARG; }
      if (e == ST_COUNT) { set_error("internal: private signal count differs from the structure pass"); return ZKR_ERR_HIP; }
      if (i > 0 && chain == i) { set_error("batch %zu: transaction %u violates: %s", bi, i, TX_STMT_TEXT[ST_CHAIN]); return ZKR_ERR_UNSATISFIED; }
      if (e) { set_error("batch %zu: transaction %u violates: %s", bi, i, TX_STMT_TEXT[e < ST_COUNT ? e : 0]); return ZKR_ERR_UNSATISFIED; }
    }
  }
  return ZKR_OK;
}

// The same value program run on the HOST for one batch (test hook: no device involved): witness_out = nVars x 32 B laid out as
// zkr_rollup_witness returns it; *stmt = the first violated statement's code (0 = none), *tx its transaction.  tests/test_rollup.py
// compares it with the host Builder byte for byte, so the program the GPU threads run is checked where no GPU exists.
int zkr_rollup_witness_program_host(uint32_t batch, uint32_t depth, const uint8_t *inputs, size_t n_inputs, uint8_t *witness_out, uint32_t *stmt, uint32_t *tx) {
  if (!inputs || !witness_out || !stmt || !tx) { set_error("null argument"); return ZKR_ERR_ARG; }
  int rc = rollup_check_geometry(batch, depth);
  if (rc) return rc;
  const uint32_t p = rollup_n_public(batch, depth), K = rollup_tx_private_count(depth);
  if (n_inputs != p - 1) { set_error("rollup circuit (%u, %u) takes %u inputs, got %zu", batch, depth, p - 1, n_inputs); return ZKR_ERR_ARG; }
  std::vector<Fr> tab(2 * 253), in(p - 1), w(K), ws(TX_WS_ELEMS), roots(batch);
  TxConsts k;
  rollup_device_constants(&k.a, &k.d, k.suborder_m1, tab.data(), tab.data() + 253);
  k.b8x = tab.data(), k.b8y = tab.data() + 253, k.mimc = mimc_round_constants();
  memcpy(in.data(), inputs, (size_t)(p - 1) * 32);
  Fr *out = reinterpret_cast<Fr *>(witness_out);
  *stmt = 0, *tx = 0;
  for (uint32_t i = 0; i < batch; i++) {
    uint32_t e = 0;  // part by part, as the GPU builder's wavefronts run them; the smallest code is the first violation in program order
    for (uint32_t part = 0; part < TX_PARTS; part++) {
      Fr r;
      const uint32_t ep = tx_witness(in.data(), batch, depth, i, K, k, w.data(), ws.data(), &r, (int)part);
      if (ep && (!e || ep < e)) e = ep;
      if (part == TX_PARTS - 1) roots[i] = r;
    }
    {
      const uint32_t ef = tx_finish(w.data(), depth, K);
      if (ef && (!e || ef < e)) e = ef;
    }
    if (i > 0) {  // the root chain comes first in circuit order (batchprocesstx.circom:67-69), as in rollup.cpp batch_witness
      Fr want;
      tx_read_input(&in[i], want);
      if (!(roots[i - 1] == want)) e = ST_CHAIN;
    }
    if (e && !*stmt) { *stmt = e; *tx = i; }
    for (uint32_t s = 0; s < K; s++) out[(size_t)p + 1 + (size_t)i * K + s] = from_mont(w[s]);
  }
  Fr one = Fr::zero();
  one.v[0] = 1;
  out[0] = one;
  out[1] = from_mont(roots[batch - 1]);
  for (uint32_t s = 2; s <= p; s++) out[s] = in[s - 2];
  return ZKR_OK;
}
const char *zkr_rollup_statement_text(uint32_t stmt) { return stmt < ST_COUNT ? TX_STMT_TEXT[stmt] : ""; }

int zkr_balance_tree_build(const void *leaves_std, unsigned depth, void *levels_out, int device) {
  if (!leaves_std || !levels_out || depth < 1 || depth > 28) { set_error("bad argument"); return ZKR_ERR_ARG; }
  if (zkr_device_count() <= device || device < 0) { set_error("no HIP device %d", device); return ZKR_ERR_NO_DEVICE; }
  ZKR_HIP_CHECK(hipSetDevice(device));
  int rc = upload_constants(device);
  if (rc) return rc;
  const size_t n = (size_t)1 << depth, total = 2 * n - 1;
  Fr *d = nullptr;
  ZKR_HIP_CHECK(hipMalloc(&d, total * 32));
  hipError_t e = hipMemcpy(d, leaves_std, n * 32, hipMemcpyHostToDevice);
  size_t off = 0, width = n;
  while (e == hipSuccess && width > 1) {  // level l+1 = hashLeftRight of adjacent pairs of level l (merkletree.ts:70-82)
    mimc_multihash_kernel<<<(unsigned)((width / 2 + 255) / 256), 256>>>(d + off, width / 2, 2, d + off + width);
    e = hipGetLastError();
    off += width;
    width >>= 1;
  }
  if (e == hipSuccess) e = hipMemcpy(levels_out, d, total * 32, hipMemcpyDeviceToHost);
  hipFree(d);
  if (e != hipSuccess) { set_error("tree build failed: %s", hipGetErrorString(e)); return ZKR_ERR_HIP; }
  return ZKR_OK;
}

}  // extern "C"
