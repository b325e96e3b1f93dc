// zkr_verify.hip -- Groth16 acceptance check (host only): the pairing equation that
// `groth.isValid(vk, proof, publicSignals)` evaluates in the reference's proof facade
// (/root/reference/operator/src/snarks/common.ts:30-38) and that TxVerifier.verify evaluates on chain
// (/root/reference/contracts/contracts/TxVerifier.sol:258-276):
//     vk_x = IC_0 + sum_i input_i * IC_{i+1};   e(-A, B) e(alfa, beta) e(vk_x, gamma) e(C, delta) == 1.
// SURVEY.md 8(a5) / 8(f-4).  No GPU involved (a few milliseconds of host time per proof).
#include <string.h>
#include <vector>
#include "pairing.hpp"
#include "zkr_internal.hpp"

using namespace zkr;

static bool fq_lt_q(const uint8_t *p) {
  uint32_t v[8];
  memcpy(v, p, 32);
  for (int i = 7; i >= 0; i--)
    if (v[i] != FqParams::P[i]) return v[i] < FqParams::P[i];
  return false;
}
static bool fr_lt_r(const uint8_t *p) {
  uint32_t v[8];
  memcpy(v, p, 32);
  for (int i = 7; i >= 0; i--)
    if (v[i] != FrParams::P[i]) return v[i] < FrParams::P[i];
  return false;
}
// standard-form bytes -> Montgomery affine; false when a coordinate is >= q or the point is off the curve
static bool read_g1(const uint8_t *p, G1Affine &out) {
  if (!fq_lt_q(p) || !fq_lt_q(p + 32)) return false;
  out = G1Affine{to_mont(load_fp<FqParams>(p)), to_mont(load_fp<FqParams>(p + 32))};
  return !out.is_inf() && pairing::g1_on_curve(out);
}
static bool read_g2(const uint8_t *p, G2Affine &out) {
  for (int i = 0; i < 4; i++)
    if (!fq_lt_q(p + 32 * i)) return false;
  out = G2Affine{Fq2{to_mont(load_fp<FqParams>(p)), to_mont(load_fp<FqParams>(p + 32))},
                 Fq2{to_mont(load_fp<FqParams>(p + 64)), to_mont(load_fp<FqParams>(p + 96))}};
  return !out.is_inf() && pairing::g2_on_curve(out);
}

extern "C" int zkr_verify(const void *vk_bin, size_t vk_len, const uint8_t proof[256], const void *public_std, size_t n_public, int *valid) {
  if (!vk_bin || !proof || !valid || (n_public && !public_std)) { set_error("null argument"); return ZKR_ERR_ARG; }
  *valid = 0;
  const uint8_t *vk = (const uint8_t *)vk_bin;
  const size_t fixed = 64 + 3 * 128 + 4;
  if (vk_len < fixed) { set_error("verifying key shorter than its fixed part (%zu bytes)", fixed); return ZKR_ERR_BAD_KEY; }
  uint32_t n_ic;
  memcpy(&n_ic, vk + 64 + 3 * 128, 4);
  if (vk_len != fixed + 64ull * n_ic) { set_error("verifying key length %zu does not match %u IC points", vk_len, n_ic); return ZKR_ERR_BAD_KEY; }
  if (n_ic != n_public + 1) { set_error("%zu public signals for a key with %u IC points (needs nPublic + 1, TxVerifier.sol:261)", n_public, n_ic); return ZKR_ERR_ARG; }
  G1Affine alfa1, ic0;
  G2Affine beta2, gamma2, delta2;
  if (!read_g1(vk, alfa1) || !read_g2(vk + 64, beta2) || !read_g2(vk + 192, gamma2) || !read_g2(vk + 320, delta2) || !read_g1(vk + fixed, ic0)) {
    set_error("verifying key holds a point that is not on the curve");
    return ZKR_ERR_BAD_KEY;
  }
  // the whole key is checked before the proof is looked at: a malformed key is an error, whatever the proof
  std::vector<G1Affine> ics(n_public);
  for (size_t i = 0; i < n_public; i++)
    if (!read_g1(vk + fixed + 64 * (i + 1), ics[i])) { set_error("verifying key IC[%zu] is not on the curve", i + 1); return ZKR_ERR_BAD_KEY; }
  // proof points: off-curve or out-of-range coordinates simply do not verify
  G1Affine a, c;
  G2Affine b;
  if (!read_g1(proof, a) || !read_g2(proof + 64, b) || !read_g1(proof + 192, c)) return 0;
  // vk_x = IC_0 + sum input_i IC_{i+1}; every input must be < r (TxVerifier.sol:265).  Interleaved 4-bit windows
  // (Straus): one shared chain of 252 doublings, one addition per input and window.
  std::vector<G1XYZZ> tab(n_public * 15);  // tab[i * 15 + d - 1] = d * IC_{i+1}
  const uint8_t *pub = (const uint8_t *)public_std;
  for (size_t i = 0; i < n_public; i++) {
    if (!fr_lt_r(pub + 32 * i)) return 0;
    G1XYZZ base = to_xyzz(ics[i]), cur = base;
    for (int d = 0; d < 15; d++) { tab[i * 15 + d] = cur; cur = add_full(cur, base); }
  }
  G1XYZZ acc = G1XYZZ::inf();
  for (int w = 63; w >= 0; w--) {
    if (w != 63) for (int k = 0; k < 4; k++) acc = dbl_xyzz(acc);
    for (size_t i = 0; i < n_public; i++) {
      unsigned d = (pub[32 * i + w / 2] >> ((w & 1) * 4)) & 15u;
      if (d) acc = add_full(acc, tab[i * 15 + d - 1]);
    }
  }
  G1XYZZ vkx = add_full(acc, to_xyzz(ic0));
  G1Affine ps[4] = {G1Affine{a.x, neg(a.y)}, alfa1, to_affine(vkx), c};
  G2Affine qs[4] = {b, beta2, gamma2, delta2};
  *valid = pairing::pairing_product_is_one(ps, qs, 4) ? 1 : 0;
  return 0;
}
