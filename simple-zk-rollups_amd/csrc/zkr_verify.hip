// zkr_verify.hip -- Groth16 acceptance check (host only): the pairing equation that
// `groth.isValid(vk, proof, publicSignals)` evaluates in the reference's proof facade
// (/root/reference/operator/src/snarks/common.ts:30-38) and that TxVerifier.verify evaluates on chain
// (/root/reference/contracts/contracts/TxVerifier.sol:258-276):
//     vk_x = IC_0 + sum_i input_i * IC_{i+1};   e(-A, B) e(alfa, beta) e(vk_x, gamma) e(C, delta) == 1.
// SURVEY.md 8(a5) / 8(f-4).  No GPU involved (a few milliseconds of host time per proof).
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <list>
#include <memory>
#include <mutex>
#include <vector>
#include "pairing.hpp"
#include <system_error>
#include <thread>
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
  // on the twist AND in the order-r subgroup (the twist has cofactor 2q - r): what precompile 8 accepts
  return !out.is_inf() && pairing::g2_on_curve(out) && pairing::g2_in_subgroup(out);
}

namespace {
struct ParsedVk {
  G1Affine alfa1, ic0;
  G2Affine beta2, gamma2, delta2;
  std::vector<G1Affine> ics;  // IC_1 .. IC_nPublic
};
// the whole key is checked before any proof is looked at: a malformed key is an error, whatever the proofs
int parse_vk(const void *vk_bin, size_t vk_len, size_t n_public, ParsedVk &k) {
  const uint8_t *vk = (const uint8_t *)vk_bin;
  const size_t fixed = 64 + 3 * 128 + 4;
  if (vk_len < fixed) { set_error("verifying key shorter than its fixed part (%zu bytes)", fixed); return ZKR_ERR_BAD_KEY; }
  uint32_t n_ic;
  memcpy(&n_ic, vk + 64 + 3 * 128, 4);
  if (vk_len != fixed + 64ull * n_ic) { set_error("verifying key length %zu does not match %u IC points", vk_len, n_ic); return ZKR_ERR_BAD_KEY; }
  if (n_ic != n_public + 1) { set_error("%zu public signals for a key with %u IC points (needs nPublic + 1, TxVerifier.sol:261)", n_public, n_ic); return ZKR_ERR_ARG; }
  if (!read_g1(vk, k.alfa1) || !read_g2(vk + 64, k.beta2) || !read_g2(vk + 192, k.gamma2) || !read_g2(vk + 320, k.delta2) || !read_g1(vk + fixed, k.ic0)) {
    set_error("verifying key holds a point that is not on the curve (or a G2 point outside the order-r subgroup)");
    return ZKR_ERR_BAD_KEY;
  }
  k.ics.resize(n_public);
  for (size_t i = 0; i < n_public; i++)
    if (!read_g1(vk + fixed + 64 * (i + 1), k.ics[i])) { set_error("verifying key IC[%zu] is not on the curve", i + 1); return ZKR_ERR_BAD_KEY; }
  return 0;
}
// sum_i scalar_i * IC_{i+1}, scalars standard 32 B each: interleaved 4-bit windows (Straus), one shared chain of 252
// doublings, one addition per scalar and window
G1XYZZ ic_combination(const ParsedVk &k, const uint8_t *scalars) {
  const size_t n = k.ics.size();
  std::vector<G1XYZZ> tab(n * 15);  // tab[i * 15 + d - 1] = d * IC_{i+1}
  for (size_t i = 0; i < n; i++) {
    G1XYZZ base = to_xyzz(k.ics[i]), cur = base;
    for (int d = 0; d < 15; d++) { tab[i * 15 + d] = cur; cur = add_full(cur, base); }
  }
  G1XYZZ acc = G1XYZZ::inf();
  for (int w = 63; w >= 0; w--) {
    if (w != 63) for (int j = 0; j < 4; j++) acc = dbl_xyzz(acc);
    for (size_t i = 0; i < n; i++) {
      unsigned d = (scalars[32 * i + w / 2] >> ((w & 1) * 4)) & 15u;
      if (d) acc = add_full(acc, tab[i * 15 + d - 1]);
    }
  }
  return acc;
}

// ---- verifying keys seen before.  The reference checks every proof against the SAME key (operator/src/snarks/common.ts:24,30:
// one vk per circuit), and a call spent most of its time on the key, not the proof: three G2 membership tests and nPublic + 2
// curve checks to parse it, then a Straus combination of its IC points from tables rebuilt per call (73 inputs for tx.circom:
// ~6 * 10^4 field multiplications, more than the four Miller loops and the final exponentiation together).  A key's bytes are
// parsed once per process; from its second use on the IC combination runs over 8-bit window tables of AFFINE multiples
// (d * IC_i, d < 256: 32 mixed additions per input instead of 64 full ones).  Entries are immutable once published (several
// host threads verify at once); the least recently used of VK_CACHE_SLOTS goes.
constexpr size_t VK_CACHE_SLOTS = 4;
struct VkEntry {
  std::vector<uint8_t> bytes;
  ParsedVk k;
  std::vector<G1Affine> win8;  // [i * 255 + d - 1] = d * IC_{i+1}; empty until the key's second use
  pairing::G2Prepared beta, gamma, delta;  // the line coefficients of the key's fixed G2 points (pairing.hpp g2_prepare), built with the entry
  pairing::Fq12 f_alfa_beta;               // Miller value of (alfa, beta): both arguments belong to the key
  int uses = 0;
};
std::mutex vk_mu;
std::list<std::shared_ptr<VkEntry>> vk_cache;

void build_win8(VkEntry &e) {
  const size_t n = e.k.ics.size();
  std::vector<G1XYZZ> pts(n * 255);
  for (size_t i = 0; i < n; i++) {
    G1XYZZ base = to_xyzz(e.k.ics[i]), cur = base;
    for (int d = 0; d < 255; d++) { pts[i * 255 + d] = cur; cur = add_mixed(cur, e.k.ics[i]); }
  }
  // affine with ONE inversion: x = X / ZZ, y = Y / ZZZ, and 1 / ZZ = ZZ^2 / ZZZ^2 (ZZ^3 = ZZZ^2), so only the ZZZ are inverted
  std::vector<Fq> pre(pts.size());
  Fq run = Fq::one();
  for (size_t j = 0; j < pts.size(); j++) {
    pre[j] = run;
    if (!pts[j].is_inf()) run = mul(run, pts[j].zzz);
  }
  Fq iv = inv(run);
  e.win8.assign(pts.size(), G1Affine{Fq::zero(), Fq::one()});
  for (size_t j = pts.size(); j-- > 0;) {
    if (pts[j].is_inf()) continue;  // d * IC = infinity cannot happen for a point of prime order r > 255; kept for safety
    Fq izzz = mul(iv, pre[j]);
    iv = mul(iv, pts[j].zzz);
    Fq izz = mul(sqr(pts[j].zz), sqr(izzz));
    e.win8[j] = G1Affine{mul(pts[j].x, izz), mul(pts[j].y, izzz)};
  }
}
// the parsed key for these bytes (ZKR_ERR_* through *rc when they are not a well-formed key for n_public inputs)
std::shared_ptr<const VkEntry> cached_vk(const void *vk_bin, size_t vk_len, size_t n_public, int *rc) {
  *rc = 0;
  std::shared_ptr<VkEntry> copy;     // ... and the private copy of it they are built on
  std::shared_ptr<VkEntry> upgrade;  // the entry whose window tables this call builds -- OUTSIDE the lock: 255 x nPublic mixed
                                     // additions (tens of ms for the ~640 inputs of a 2^20 rollup key) must not stall the other verifiers
  {
    std::lock_guard<std::mutex> lk(vk_mu);
    for (auto it = vk_cache.begin(); it != vk_cache.end(); ++it) {
      if ((*it)->bytes.size() != vk_len || memcmp((*it)->bytes.data(), vk_bin, vk_len) != 0) continue;
      std::shared_ptr<VkEntry> e = *it;
      if (e->k.ics.size() != n_public) break;  // the count check below words the error
      vk_cache.splice(vk_cache.begin(), vk_cache, it);
      if (++e->uses != 2 || !e->win8.empty()) return e;
      upgrade = e;  // second use: worth the tables (uses is past 2 now: no second caller starts the same build)
      copy = std::make_shared<VkEntry>(*e);  // the copy is taken UNDER the lock: other verifiers bump e->uses under it meanwhile
      break;
    }
  }
  if (upgrade) {
    std::shared_ptr<VkEntry> e2 = copy;
    build_win8(*e2);
    std::lock_guard<std::mutex> lk(vk_mu);  // publish: the new entry replaces the old one where it still sits (readers keep theirs)
    for (auto &slot : vk_cache)
      if (slot == upgrade) { e2->uses = upgrade->uses; slot = e2; break; }
    return e2;
  }
  auto e = std::make_shared<VkEntry>();
  if ((*rc = parse_vk(vk_bin, vk_len, n_public, e->k))) return nullptr;
  if (!pairing::g2_prepare(e->k.beta2, e->beta) || !pairing::g2_prepare(e->k.gamma2, e->gamma) || !pairing::g2_prepare(e->k.delta2, e->delta)) {
    set_error("verifying key: a G2 point is not a valid pairing input");  // an undefined slope: impossible for points of order r, which parse_vk checks
    *rc = ZKR_ERR_BAD_KEY;
    return nullptr;
  }
  {
    const pairing::G2Prepared *qb = &e->beta;
    e->f_alfa_beta = pairing::miller_loop_mixed(nullptr, nullptr, 0, &e->k.alfa1, &qb, 1);
  }
  e->bytes.assign((const uint8_t *)vk_bin, (const uint8_t *)vk_bin + vk_len);
  e->uses = 1;
  std::lock_guard<std::mutex> lk(vk_mu);
  vk_cache.push_front(e);
  while (vk_cache.size() > VK_CACHE_SLOTS) vk_cache.pop_back();
  return e;
}
// sum_i scalar_i * IC_{i+1} for a cached key: the window tables when it has them, the per-call Straus tables otherwise
G1XYZZ ic_combination(const VkEntry &e, const uint8_t *scalars) {
  if (e.win8.empty()) return ic_combination(e.k, scalars);
  const size_t n = e.k.ics.size();
  G1XYZZ acc = G1XYZZ::inf();
  for (int w = 31; w >= 0; w--) {
    if (w != 31) for (int j = 0; j < 8; j++) acc = dbl_xyzz(acc);
    for (size_t i = 0; i < n; i++) {
      const unsigned d = scalars[32 * i + w];
      if (d) acc = add_mixed(acc, e.win8[i * 255 + d - 1]);
    }
  }
  return acc;
}
}  // namespace

extern "C" int zkr_verify(const void *vk_bin, size_t vk_len, const uint8_t proof[256], const void *public_std, size_t n_public, int *valid) {
  if (!vk_bin || !proof || !valid || (n_public && !public_std)) { set_error("null argument"); return ZKR_ERR_ARG; }
  *valid = 0;
  int rc = 0;
  std::shared_ptr<const VkEntry> ent = cached_vk(vk_bin, vk_len, n_public, &rc);
  if (!ent) return rc;
  const ParsedVk &k = ent->k;
  // proof points: off-curve or out-of-range coordinates simply do not verify
  G1Affine a, c;
  G2Affine b;
  if (!read_g1(proof, a) || !read_g1(proof + 192, c)) return 0;
  // vk_x = IC_0 + sum input_i IC_{i+1}; every input must be < r (TxVerifier.sol:265)
  const uint8_t *pub = (const uint8_t *)public_std;
  for (size_t i = 0; i < n_public; i++)
    if (!fr_lt_r(pub + 32 * i)) return 0;
  // e(-A, B) e(alfa, beta) e(vk_x, gamma) e(C, delta) == 1 in two halves that share nothing until the end: a helper thread takes
  // the public inputs -- vk_x from the window tables (a third of the whole check) and the Miller loop of (vk_x, gamma) over gamma's
  // prepared lines --, this thread the proof's own points -- B's membership test, then (-A, B) by projective steps with (C, delta)
  // over delta's prepared lines; the Miller value of (alfa, beta) comes with the key.  1.5 -> ~0.9 ms per isValid on the GPU box's
  // host; ZKR_VERIFY_THREADS=1: one thread.
  static const bool one_thread = getenv("ZKR_VERIFY_THREADS") && atoi(getenv("ZKR_VERIFY_THREADS")) <= 1;
  pairing::Fq12 f_pub = pairing::Fq12::one();
  auto public_half = [&]() {
    const G1XYZZ vkx = add_full(ic_combination(*ent, pub), to_xyzz(k.ic0));
    const G1Affine px = to_affine(vkx);
    const pairing::G2Prepared *qg = &ent->gamma;
    f_pub = pairing::miller_loop_mixed(nullptr, nullptr, 0, &px, &qg, 1);
  };
  std::thread helper;
  if (one_thread) public_half();
  else {
    // no thread to be had (EAGAIN under a pids / thread limit: many libuv or Python workers verifying at once) must not
    // terminate the host process through the C ABI: the public half then runs here
    try { helper = std::thread(public_half); } catch (const std::system_error &) { public_half(); }
  }
  bool ok = read_g2(proof + 64, b);  // on the twist and in G2
  pairing::Fq12 f_proof = pairing::Fq12::one();
  if (ok) {
    const G1Affine pa{a.x, neg(a.y)};
    const pairing::G2Prepared *qd = &ent->delta;
    f_proof = pairing::miller_loop_mixed(&pa, &b, 1, &c, &qd, 1, &ok);
  }
  if (helper.joinable()) helper.join();
  if (!ok) return 0;
  const pairing::Fq12 f = pairing::mul(pairing::mul(ent->f_alfa_beta, f_pub), f_proof);
  *valid = pairing::final_exponentiation(f) == pairing::Fq12::one() ? 1 : 0;
  return 0;
}

// Many proofs under one key with ONE final exponentiation: each proof's equation is raised to a random 128-bit z_i
// (z_0 = 1; OS CSPRNG) and the products are merged,
//     prod_i e(-z_i A_i, B_i) * e((sum z_i) alfa, beta) * e(sum_i z_i vk_x_i, gamma) * e(sum_i z_i C_i, delta) == 1,
// where sum_i z_i vk_x_i = (sum z_i) IC_0 + sum_j (sum_i z_i input_ij mod r) IC_{j+1} is one combination of the IC points.
// n + 3 Miller loops instead of 4 n; a batch with an invalid proof passes with probability 2^-128.
extern "C" int zkr_verify_batch(const void *vk_bin, size_t vk_len, const uint8_t *proofs, const void *publics_std, size_t n_proofs, size_t n_public, int *all_valid) {
  if (!vk_bin || !all_valid || (n_proofs && (!proofs || (n_public && !publics_std)))) { set_error("null argument"); return ZKR_ERR_ARG; }
  *all_valid = 0;
  int rc = 0;
  std::shared_ptr<const VkEntry> ent = cached_vk(vk_bin, vk_len, n_public, &rc);
  if (!ent) return rc;
  const ParsedVk &k = ent->k;
  if (n_proofs == 0) { *all_valid = 1; return 0; }
  const uint8_t *pub = (const uint8_t *)publics_std;
  std::vector<uint8_t> z(32 * n_proofs, 0);
  {
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f) { set_error("cannot open /dev/urandom"); return ZKR_ERR_ARG; }
    bool bad = false;
    for (size_t i = 1; i < n_proofs && !bad; i++) bad = fread(&z[32 * i], 1, 16, f) != 16;
    fclose(f);
    if (bad) { set_error("short read from /dev/urandom"); return ZKR_ERR_ARG; }
    z[0] = 1;
  }
  std::vector<G1Affine> ps;
  std::vector<G2Affine> qs;
  std::vector<Fr> comb(n_public, Fr::zero());  // sum_i z_i input_ij (Montgomery)
  Fr zsum = Fr::zero();
  G1XYZZ csum = G1XYZZ::inf();
  for (size_t i = 0; i < n_proofs; i++) {
    const uint8_t *pr = proofs + 256 * i;
    G1Affine a, c;
    G2Affine b;
    if (!read_g1(pr, a) || !read_g2(pr + 64, b) || !read_g1(pr + 192, c)) return 0;
    U256 zi = load_u256(&z[32 * i]);
    Fr zm = to_mont(load_fp<FrParams>(&z[32 * i]));
    zsum = add(zsum, zm);
    for (size_t j = 0; j < n_public; j++) {
      const uint8_t *v = pub + 32 * (i * n_public + j);
      if (!fr_lt_r(v)) return 0;
      comb[j] = add(comb[j], mul(zm, to_mont(load_fp<FrParams>(v))));
    }
    G1XYZZ za = scalar_mul(to_xyzz(G1Affine{a.x, neg(a.y)}), zi);  // 128-bit scalars: the loop skips the leading zero bits cheaply
    csum = add_full(csum, scalar_mul(to_xyzz(c), zi));
    ps.push_back(to_affine(za));
    qs.push_back(b);
  }
  std::vector<uint8_t> comb_std(32 * n_public);
  for (size_t j = 0; j < n_public; j++) store_fp(&comb_std[32 * j], from_mont(comb[j]));
  Fr zs_std = from_mont(zsum);
  U256 zs;
  memcpy(zs.v, zs_std.v, 32);
  G1XYZZ vkx = add_full(ic_combination(*ent, comb_std.data()), scalar_mul(to_xyzz(k.ic0), zs));
  const G1Affine pf[3] = {to_affine(scalar_mul(to_xyzz(k.alfa1), zs)), to_affine(vkx), to_affine(csum)};
  const pairing::G2Prepared *qf[3] = {&ent->beta, &ent->gamma, &ent->delta};
  bool ok = true;  // one loop over all pairs: the proofs' B_i by projective steps (one squaring of f per bit for the whole batch)
  const pairing::Fq12 f = pairing::miller_loop_mixed(ps.data(), qs.data(), ps.size(), pf, qf, 3, &ok);
  if (!ok) return 0;  // a degenerate step: cannot happen for subgroup points (read_g2 checks), never "valid"
  *all_valid = pairing::final_exponentiation(f) == pairing::Fq12::one() ? 1 : 0;
  return 0;
}
