// rollup_gpu.hip -- MiMCSponge-220 on the GPU, batch-parallel: the hash of the reference's balance tree and leaves
// (/root/reference/operator/src/utils/crypto.ts:28-38 `multiHash` / `hashLeftRight`, prover/circuits/hasher.circom:3-30;
// operator/src/utils/merkletree.ts:44-83 builds the tree with it).  SURVEY 8(f-3): "batch-parallel ... on GPU" -- one thread
// per hash, 220 rounds of x -> x^5 in Fr Montgomery arithmetic (3 multiplications per round), so a tree of 2^20 accounts
// (2^21 - 1 hashes of two elements) is 2.8 * 10^9 multiplications: tens of milliseconds here, minutes in the circomlib
// BigInt code.  Host counterpart and round constants: rollup.cpp; parity against oracle/rollup.py in tests/test_gpu_rollup.py.
#include "zkr_internal.hpp"

namespace zkr {
const Fr *mimc_round_constants();  // rollup.cpp

constexpr int MIMC_ROUNDS = 220;
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
