// workload.hip -- seeded synthetic rollup-shaped R1CS, witness, and Groth16 setup (SURVEY.md 8(d)
// configs 2-5, 8(f-2)).  Host logic here; the key's group elements are computed on the GPU by the
// fixed-base kernel (kernels_msm.hpp), so no multi-GB websnark binary ever exists on the host.
//
// The reference has no checked-in key or circuit artefact (prover/.gitignore:109), so benchmark
// inputs must be generated.  Restates `snarkjs setup --protocol groth` (prover/package.json:34,37;
// SURVEY App. B "Setup") for a circuit drawn by the same SplitMix64 stream as oracle/groth16.py's
// synth_circuit -- tests/test_workload.py checks the two generators byte-for-byte at small sizes.
#include <stdlib.h>
#include <string.h>
#include <map>
#include "zkr_internal.hpp"

namespace zkr {

struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t u64() {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  Fr fr_std() {  // uniform in [0, r) by rejection, standard form
    for (;;) {
      uint64_t a = u64(), b = u64(), c = u64(), d = u64() & ((1ull << 62) - 1);
      Fr v;
      v.v[0] = (uint32_t)a; v.v[1] = (uint32_t)(a >> 32); v.v[2] = (uint32_t)b; v.v[3] = (uint32_t)(b >> 32);
      v.v[4] = (uint32_t)c; v.v[5] = (uint32_t)(c >> 32); v.v[6] = (uint32_t)d; v.v[7] = (uint32_t)(d >> 32);
      bool lt = false;
      for (int i = 7; i >= 0; i--)
        if (v.v[i] != FrParams::P[i]) { lt = v.v[i] < FrParams::P[i]; break; }
      if (lt) return v;
    }
  }
  Fr fr() { return to_mont(fr_std()); }
};

static Fr fr_u64(uint64_t x) {
  Fr r = Fr::zero();
  r.v[0] = (uint32_t)x;
  r.v[1] = (uint32_t)(x >> 32);
  return to_mont(r);
}

struct Term { uint32_t sig; Fr coef; };  // coef Montgomery
struct Circuit {
  uint32_t n = 0, p = 0, nC = 0, m = 0;
  std::vector<uint32_t> rowA, rowB, rowC;  // CSR row pointers (nC+1)
  std::vector<Term> tA, tB, tC;
  std::vector<Fr> w;  // Montgomery
};

static Fr dot(const std::vector<Term> &t, size_t b, size_t e, const std::vector<Fr> &w) {
  Fr acc = Fr::zero();
  for (size_t i = b; i < e; i++) acc = add(acc, mul(t[i].coef, w[t[i].sig]));
  return acc;
}

// circuit shape of the zkr_synth_* entry points: 0 = rollup-shaped (default), 1 = dense random (BASELINE config 5)
static thread_local unsigned g_shape = 0;  // per calling thread (zkr_synth_set_shape): the library stays re-entrant

// draw-for-draw mirror of oracle/groth16.py:synth_circuit
static void synth_circuit(Circuit &c, uint32_t m, uint32_t p, uint64_t seed, uint64_t witness_seed) {
  SplitMix64 rng(seed);
  SplitMix64 rv(witness_seed ^ 0x77697473ull);  // free witness values: own stream, so one key serves many witnesses
  c.m = m; c.p = p; c.nC = m - p - 1;
  c.w.reserve(m + 8);
  c.w.push_back(Fr::one());
  for (uint32_t i = 0; i < p; i++) c.w.push_back(rv.fr());
  const Fr one = Fr::one(), minus1 = neg(Fr::one());
  c.rowA.push_back(0); c.rowB.push_back(0); c.rowC.push_back(0);
  for (uint32_t row = 0; row < c.nC; row++) {
    uint32_t n = (uint32_t)c.w.size();
    if (g_shape == 1) {  // (4 random signals, random coefficients) x (4 more) = new signal
      std::map<uint32_t, Fr> AB[2];
      for (auto &d : AB)
        for (int k = 0; k < 4; k++) {
          uint32_t j = (uint32_t)(rng.u64() % n);
          Fr cf = rng.fr();
          auto it = d.find(j);
          if (it == d.end()) d[j] = cf; else it->second = add(it->second, cf);
        }
      size_t ba = c.tA.size(), bb = c.tB.size();
      for (auto &kv : AB[0]) c.tA.push_back({kv.first, kv.second});
      for (auto &kv : AB[1]) c.tB.push_back({kv.first, kv.second});
      c.w.push_back(mul(dot(c.tA, ba, c.tA.size(), c.w), dot(c.tB, bb, c.tB.size(), c.w)));
      c.tC.push_back({n, one});
      c.rowA.push_back((uint32_t)c.tA.size());
      c.rowB.push_back((uint32_t)c.tB.size());
      c.rowC.push_back((uint32_t)c.tC.size());
      continue;
    }
    uint64_t kind = rng.u64() % 100;
    if (row % 2048 == 1000) {
      std::map<uint32_t, Fr> A;
      Fr pw = one;
      for (int k = 0; k < 64; k++) {
        uint32_t j = (uint32_t)(rng.u64() % n);
        auto it = A.find(j);
        if (it == A.end()) A[j] = pw; else it->second = add(it->second, pw);
        pw = dbl(pw);
      }
      size_t b = c.tA.size();
      for (auto &kv : A) c.tA.push_back({kv.first, kv.second});
      c.w.push_back(dot(c.tA, b, c.tA.size(), c.w));
      c.tB.push_back({0, one});
      c.tC.push_back({n, one});
    } else if (kind < 3) {
      uint64_t bit = rv.u64() & 1;
      c.w.push_back(bit ? one : Fr::zero());
      c.tA.push_back({n, one});
      c.tB.push_back({0, minus1});
      c.tB.push_back({n, one});
    } else if (kind < 5) {
      c.w.push_back(fr_u64(rv.u64()));
      c.tA.push_back({n, one});
      c.tB.push_back({0, one});
      c.tC.push_back({n, one});
    } else {
      uint32_t i = n - 1;
      uint64_t sel = rng.u64();
      uint32_t j = (uint32_t)(rng.u64() % n);
      uint32_t kk = 0;
      if (n > 1) { uint32_t lim = n - 1 < 16 ? n - 1 : 16; kk = n - 1 - (uint32_t)(rng.u64() % lim); }
      std::map<uint32_t, Fr> A, B;
      A[i] = one;
      if ((sel & 1) && j != i) A[j] = rng.fr();
      if (((sel >> 1) & 3) == 0 && !A.count(0)) A[0] = rng.fr();
      B[kk] = one;
      if (((sel >> 3) & 1) && kk != 0) B[0] = rng.fr();
      size_t ba = c.tA.size(), bb = c.tB.size();
      for (auto &kv : A) c.tA.push_back({kv.first, kv.second});
      for (auto &kv : B) c.tB.push_back({kv.first, kv.second});
      Fr val = mul(dot(c.tA, ba, c.tA.size(), c.w), dot(c.tB, bb, c.tB.size(), c.w));
      if (((sel >> 4) & 3) == 0) {
        uint32_t i2 = (uint32_t)(rng.u64() % n);
        c.tC.push_back({i2, minus1});
        val = add(val, c.w[i2]);
      }
      c.tC.push_back({n, one});
      c.w.push_back(val);
    }
    c.rowA.push_back((uint32_t)c.tA.size());
    c.rowB.push_back((uint32_t)c.tB.size());
    c.rowC.push_back((uint32_t)c.tC.size());
  }
  c.n = (uint32_t)c.w.size();
}

struct Toxic { Fr t, alfa, beta, gamma, delta; };  // Montgomery

struct SetupScalars {
  std::vector<Fr> a, b, c;            // per signal, Montgomery
  std::vector<Fr> cpriv, ic, hx;      // (beta a + alfa b + c)/delta for s>p ; /gamma for s<=p ; t^i Z(t)/delta
};

Fr host_root_of_unity(unsigned k);

template <class T> static void wipe_vector(std::vector<T> &v);
static void setup_scalars(const Circuit &c, const Toxic &tx, SetupScalars &sc, bool secret) {
  uint32_t m = c.m, n = c.n, p = c.p;
  unsigned logm = 0;
  while ((1u << logm) < m) logm++;
  // L_c(t) = (t^m - 1) w^c / (m (t - w^c)) with one batch inversion
  Fr w = host_root_of_unity(logm);
  std::vector<Fr> wc(m), den(m), pref(m);
  Fr cur = Fr::one();
  for (uint32_t i = 0; i < m; i++) { wc[i] = cur; den[i] = sub(tx.t, cur); cur = mul(cur, w); }
  Fr run = Fr::one();
  for (uint32_t i = 0; i < m; i++) { pref[i] = run; run = mul(run, den[i]); }
  Fr invall = inv(run);
  Fr tm = tx.t;
  for (unsigned i = 0; i < logm; i++) tm = sqr(tm);
  Fr z = sub(tm, Fr::one());
  Fr zm = mul(z, inv(fr_u64(m)));
  std::vector<Fr> L(m);
  for (uint32_t i = m; i-- > 0;) {
    Fr di = mul(invall, pref[i]);
    invall = mul(invall, den[i]);
    L[i] = mul(mul(zm, wc[i]), di);
  }
  sc.a.assign(n, Fr::zero()); sc.b.assign(n, Fr::zero()); sc.c.assign(n, Fr::zero());
  for (uint32_t r = 0; r < c.nC; r++) {
    for (uint32_t k = c.rowA[r]; k < c.rowA[r + 1]; k++) sc.a[c.tA[k].sig] = add(sc.a[c.tA[k].sig], mul(c.tA[k].coef, L[r]));
    for (uint32_t k = c.rowB[r]; k < c.rowB[r + 1]; k++) sc.b[c.tB[k].sig] = add(sc.b[c.tB[k].sig], mul(c.tB[k].coef, L[r]));
    for (uint32_t k = c.rowC[r]; k < c.rowC[r + 1]; k++) sc.c[c.tC[k].sig] = add(sc.c[c.tC[k].sig], mul(c.tC[k].coef, L[r]));
  }
  for (uint32_t i = 0; i <= p; i++) sc.a[i] = add(sc.a[i], L[c.nC + i]);  // input-consistency rows polsA[i][nC+i] = 1
  Fr ginv = inv(tx.gamma), dinv = inv(tx.delta);
  sc.ic.resize(p + 1);
  sc.cpriv.resize(n - p - 1);
  for (uint32_t s = 0; s < n; s++) {
    Fr k = add(add(mul(tx.beta, sc.a[s]), mul(tx.alfa, sc.b[s])), sc.c[s]);
    if (s <= p) sc.ic[s] = mul(k, ginv); else sc.cpriv[s - p - 1] = mul(k, dinv);
  }
  sc.hx.resize(m);
  Fr ti = mul(z, dinv);
  for (uint32_t i = 0; i < m; i++) { sc.hx[i] = ti; ti = mul(ti, tx.t); }
  if (secret) { wipe_vector(den); wipe_vector(pref); wipe_vector(L); }  // t - w^c, their products, L_c(t): each reveals t
}

static void to_std_bytes(const std::vector<Fr> &v, std::vector<uint8_t> &out) {
  out.resize(v.size() * 32);
  for (size_t i = 0; i < v.size(); i++) { Fr s = from_mont(v[i]); memcpy(&out[i * 32], s.v, 32); }
}

template <class T>
static void wipe_vector(std::vector<T> &v) {
  if (!v.empty()) explicit_bzero((void *)v.data(), v.size() * sizeof(T));
}

struct Generated {
  Circuit circ;
  Toxic tox;
  SetupScalars sc;
  bool secret = false;  // fresh toxic waste (zkr_setup_r1cs without injected scalars): everything derived from it is wiped
  void *d_tbl[N_TABLES] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  uint8_t consts[448];
  std::vector<uint8_t> ic_std, gamma2_std;
  ~Generated() {
    for (auto p : d_tbl) if (p) hipFree(p);
    if (secret) {  // t, alfa, beta, gamma, delta and the per-signal / per-power scalars (a_s, b_s, c_s, K_s/delta, K_s/gamma, t^i Z/delta):
                   // any of them lets its holder forge proofs for the generated key
      explicit_bzero((void *)&tox, sizeof(tox));
      wipe_vector(sc.a); wipe_vector(sc.b); wipe_vector(sc.c); wipe_vector(sc.cpriv); wipe_vector(sc.ic); wipe_vector(sc.hx);
    }
  }
};

// g.circ and g.tox are in place: QAP polynomials at t, then every group element of the key on the GPU
static int setup_from_circuit(int device, Generated &g) {
  if (zkr_device_count() <= device || device < 0) { set_error("no HIP device %d; key points are computed on the GPU (no CPU fallback)", device); return ZKR_ERR_NO_DEVICE; }
  setup_scalars(g.circ, g.tox, g.sc, g.secret);
  struct Staging {  // standard-form copies of the scalars on their way to the device: wiped with the rest
    std::vector<uint8_t> v;
    bool secret;
    ~Staging() { if (secret) wipe_vector(v); }
  } stage{{}, g.secret};
  std::vector<uint8_t> &bytes = stage.v;
  int rc;
  const std::vector<Fr> *src[N_TABLES] = {&g.sc.a, &g.sc.b, &g.sc.b, &g.sc.cpriv, &g.sc.hx};
  for (int t = 0; t < N_TABLES; t++) {
    if (g.secret) wipe_vector(bytes);
    to_std_bytes(*src[t], bytes);
    if ((rc = fixed_base_points(device, t == T_B2, bytes.data(), src[t]->size(), &g.d_tbl[t], g.secret))) return rc;
  }
  // vk_alfa_1, vk_beta_1, vk_delta_1 | vk_beta_2, vk_delta_2  (header layout binarify.ts:163-167)
  std::vector<Fr> g1s = {g.tox.alfa, g.tox.beta, g.tox.delta}, g2s = {g.tox.beta, g.tox.delta, g.tox.gamma};
  void *d = nullptr;
  if (g.secret) wipe_vector(bytes);
  to_std_bytes(g1s, bytes);
  if ((rc = fixed_base_points(device, false, bytes.data(), 3, &d, g.secret))) return rc;
  ZKR_HIP_CHECK(hipMemcpy(g.consts, d, 192, hipMemcpyDeviceToHost));
  hipFree(d);
  to_std_bytes(g2s, bytes);
  if ((rc = fixed_base_points(device, true, bytes.data(), 3, &d, g.secret))) return rc;
  if (g.secret) { wipe_vector(g1s); wipe_vector(g2s); }
  uint8_t g2pts[384];
  ZKR_HIP_CHECK(hipMemcpy(g2pts, d, 384, hipMemcpyDeviceToHost));
  hipFree(d);
  memcpy(g.consts + 192, g2pts, 256);
  G2Affine gm = load_g2(g2pts + 256);
  g.gamma2_std.resize(128);
  store_g2_std(g.gamma2_std.data(), gm);
  // IC points for the checker
  if (g.secret) wipe_vector(bytes);
  to_std_bytes(g.sc.ic, bytes);
  if ((rc = fixed_base_points(device, false, bytes.data(), g.sc.ic.size(), &d, g.secret))) return rc;
  std::vector<uint8_t> icm(g.sc.ic.size() * 64);
  ZKR_HIP_CHECK(hipMemcpy(icm.data(), d, icm.size(), hipMemcpyDeviceToHost));
  hipFree(d);
  g.ic_std.resize(icm.size());
  for (size_t i = 0; i < g.sc.ic.size(); i++) store_g1_std(&g.ic_std[i * 64], load_g1(&icm[i * 64]));
  return 0;
}

static int generate(unsigned log_m, unsigned n_public, uint64_t circuit_seed, uint64_t toxic_seed, int device, Generated &g) {
  if (log_m < 2 || log_m > 26 || n_public + 2 > (1u << log_m)) { set_error("bad synthetic geometry log_m=%u nPublic=%u", log_m, n_public); return ZKR_ERR_ARG; }
  if (zkr_device_count() <= device || device < 0) { set_error("no HIP device %d; key points are computed on the GPU (no CPU fallback)", device); return ZKR_ERR_NO_DEVICE; }
  synth_circuit(g.circ, 1u << log_m, n_public, circuit_seed, circuit_seed);
  SplitMix64 rng(toxic_seed);
  g.tox.t = rng.fr(); g.tox.alfa = rng.fr(); g.tox.beta = rng.fr(); g.tox.gamma = rng.fr(); g.tox.delta = rng.fr();
  return setup_from_circuit(device, g);
}

static uint32_t bitrev32(uint32_t x, unsigned bits) {
  uint32_t r = 0;
  for (unsigned b = 0; b < bits; b++) r |= ((x >> b) & 1) << (bits - 1 - b);
  return r;
}

// CSR rows of the QAP (A gets the nPublic+1 input-consistency rows), the kept points of every table, then the arena
static int build_key_from_generated(const Generated &g, int device, zkr_key **key_out) {
  const Circuit &c = g.circ;
  uint32_t n = c.n, p = c.p, m = c.m;
  // QAP rows in CSR (A gets the nPublic+1 input-consistency rows)
  std::vector<uint32_t> rowptr[2], col[2];
  std::vector<uint8_t> coef[2];
  const std::vector<uint32_t> *rp[2] = {&c.rowA, &c.rowB};
  const std::vector<Term> *tt[2] = {&c.tA, &c.tB};
  for (int s = 0; s < 2; s++) {
    rowptr[s].assign(m + 1, 0);
    size_t extra = s == 0 ? p + 1 : 0;
    col[s].reserve(tt[s]->size() + extra);
    coef[s].reserve((tt[s]->size() + extra) * 32);
    for (uint32_t r = 0; r < m; r++) {
      rowptr[s][r] = (uint32_t)col[s].size();
      if (r < c.nC) {
        for (uint32_t k = (*rp[s])[r]; k < (*rp[s])[r + 1]; k++) {
          col[s].push_back((*tt[s])[k].sig);
          const uint8_t *b = (const uint8_t *)(*tt[s])[k].coef.v;
          coef[s].insert(coef[s].end(), b, b + 32);
        }
      } else if (s == 0 && r - c.nC <= p) {
        col[s].push_back(r - c.nC);
        Fr one = Fr::one();
        const uint8_t *b = (const uint8_t *)one.v;
        coef[s].insert(coef[s].end(), b, b + 32);
      }
    }
    rowptr[s][m] = (uint32_t)col[s].size();
  }
  std::vector<uint32_t> srcidx[N_TABLES], sidx[N_TABLES];
  for (uint32_t s = 0; s < n; s++) {
    if (!g.sc.a[s].is_zero()) { srcidx[T_A].push_back(s); sidx[T_A].push_back(s); }
    if (!g.sc.b[s].is_zero()) { srcidx[T_B1].push_back(s); sidx[T_B1].push_back(s); srcidx[T_B2].push_back(s); sidx[T_B2].push_back(s); }
  }
  for (uint32_t i = 0; i < n - p - 1; i++)
    if (!g.sc.cpriv[i].is_zero()) { srcidx[T_C].push_back(i); sidx[T_C].push_back(i + p + 1); }
  unsigned logm = 0;
  while ((1u << logm) < m) logm++;
  for (uint32_t j = 0; j < m; j++) {
    uint32_t i = bitrev32(j, logm);
    if (!g.sc.hx[i].is_zero()) { srcidx[T_H].push_back(i); sidx[T_H].push_back(j); }
  }
  const void *src[N_TABLES];
  bool on_dev[N_TABLES];
  for (int t = 0; t < N_TABLES; t++) { src[t] = g.d_tbl[t]; on_dev[t] = true; }
  return key_build(device, n, p, m, rowptr, col, coef, src, on_dev, srcidx, sidx, g.consts, key_out);
}

// vk_bin (zkr_verify layout) from the setup's own data
static void vk_from_generated(const Generated &g, std::vector<uint8_t> &vk) {
  const size_t nic = g.sc.ic.size();
  vk.resize(64 + 3 * 128 + 4 + 64 * nic);
  store_g1_std(vk.data(), load_g1(g.consts));
  store_g2_std(vk.data() + 64, load_g2(g.consts + 192));
  memcpy(vk.data() + 192, g.gamma2_std.data(), 128);
  store_g2_std(vk.data() + 320, load_g2(g.consts + 320));
  uint32_t nic32 = (uint32_t)nic;
  memcpy(vk.data() + 448, &nic32, 4);
  memcpy(vk.data() + 452, g.ic_std.data(), 64 * nic);
}

}  // namespace zkr

using namespace zkr;

extern "C" {

// ---- Groth16 setup for an arbitrary R1CS: restates `snarkjs setup --protocol groth`
// (/root/reference/prover/package.json:34,37; SURVEY App. B "Setup") with the key's group elements computed on the GPU
static bool read_fr_std(const uint8_t *p, Fr &out) {
  memcpy(out.v, p, 32);
  for (int i = 7; i >= 0; i--)
    if (out.v[i] != FrParams::P[i]) return out.v[i] < FrParams::P[i];
  return false;
}
static int draw_fr(FILE *f, Fr &out) {
  for (;;) {
    uint8_t b[32];
    if (fread(b, 1, 32, f) != 32) return 1;
    b[31] &= 0x3f;
    if (read_fr_std(b, out) && !out.is_zero()) return 0;
  }
}

// r1cs_bin (include/zkr.h) -> g.circ; toxic waste injected or drawn; QAP at t and every group element on the GPU
static int setup_parse_and_run(const void *r1cs_bin, size_t r1cs_len, const uint8_t *toxic160, int device, Generated &g) {
  const uint8_t *b = (const uint8_t *)r1cs_bin, *end = b + r1cs_len;
  if (r1cs_len < 12) { set_error("R1CS shorter than its header"); return ZKR_ERR_ARG; }
  Circuit &c = g.circ;
  memcpy(&c.n, b, 4); memcpy(&c.p, b + 4, 4); memcpy(&c.nC, b + 8, 4);
  b += 12;
  if (c.n < 1 || (uint64_t)c.p + 1 > c.n || c.nC < 1 || (uint64_t)c.nC + c.p + 1 > (1ull << 26)) { set_error("bad R1CS geometry nVars=%u nPublic=%u nConstraints=%u", c.n, c.p, c.nC); return ZKR_ERR_ARG; }
  c.m = 2;
  while (c.m < c.nC + c.p + 1) c.m <<= 1;  // snarkjs: domainBits = floor(log2(nC + nPublic)) + 1
  std::vector<uint32_t> *rows[3] = {&c.rowA, &c.rowB, &c.rowC};
  std::vector<Term> *terms[3] = {&c.tA, &c.tB, &c.tC};
  for (auto r : rows) r->push_back(0);
  for (uint32_t row = 0; row < c.nC; row++) {
    for (int s = 0; s < 3; s++) {
      if (b + 4 > end) { set_error("R1CS truncated in constraint %u", row); return ZKR_ERR_ARG; }
      uint32_t k;
      memcpy(&k, b, 4);
      b += 4;
      if ((uint64_t)k * 36 > (uint64_t)(end - b)) { set_error("R1CS truncated in constraint %u", row); return ZKR_ERR_ARG; }
      for (uint32_t e = 0; e < k; e++, b += 36) {
        Term t;
        memcpy(&t.sig, b, 4);
        Fr cf;
        if (t.sig >= c.n || !read_fr_std(b + 4, cf)) { set_error("constraint %u: signal %u out of range or coefficient >= r", row, t.sig); return ZKR_ERR_ARG; }
        t.coef = to_mont(cf);
        terms[s]->push_back(t);
      }
      rows[s]->push_back((uint32_t)terms[s]->size());
    }
  }
  if (b != end) { set_error("R1CS has %zu trailing bytes", (size_t)(end - b)); return ZKR_ERR_ARG; }
  Fr *tox[5] = {&g.tox.t, &g.tox.alfa, &g.tox.beta, &g.tox.gamma, &g.tox.delta};
  if (toxic160) {
    for (int i = 0; i < 5; i++) {
      Fr v;
      if (!read_fr_std(toxic160 + 32 * i, v) || v.is_zero()) { set_error("toxic scalar %d is zero or >= r", i); return ZKR_ERR_ARG; }
      *tox[i] = to_mont(v);
    }
  } else {  // fresh toxic waste from the OS CSPRNG; it never leaves this call (Generated::~Generated wipes it and
            // everything derived from it, host and device copies)
    g.secret = true;
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f) { set_error("cannot open /dev/urandom"); return ZKR_ERR_ARG; }
    int bad = 0;
    for (int i = 0; i < 5 && !bad; i++) { Fr v; bad = draw_fr(f, v); *tox[i] = to_mont(v); }
    fclose(f);
    if (bad) { set_error("short read from /dev/urandom"); return ZKR_ERR_ARG; }
  }
  return setup_from_circuit(device, g);
}

static void vk_to_malloc(const Generated &g, void **vk_out, size_t *vk_len) {
  std::vector<uint8_t> vk;
  vk_from_generated(g, vk);
  *vk_out = malloc(vk.size());
  memcpy(*vk_out, vk.data(), vk.size());
  *vk_len = vk.size();
}

int zkr_setup_r1cs(const void *r1cs_bin, size_t r1cs_len, const uint8_t *toxic160, int device, zkr_key **key_out, void **vk_out, size_t *vk_len) {
  if (!r1cs_bin || !key_out || !vk_out || !vk_len) { set_error("null argument"); return ZKR_ERR_ARG; }
  Generated g;
  int rc = setup_parse_and_run(r1cs_bin, r1cs_len, toxic160, device, g);
  if (!rc) rc = build_key_from_generated(g, device, key_out);
  if (rc) return rc;
  vk_to_malloc(g, vk_out, vk_len);
  return 0;
}

static int render_websnark(const Generated &g, void **pk_out, size_t *pk_len);

int zkr_setup_r1cs_websnark(const void *r1cs_bin, size_t r1cs_len, const uint8_t *toxic160, int device, void **pk_out, size_t *pk_len, void **vk_out, size_t *vk_len) {
  if (!r1cs_bin || !pk_out || !pk_len || !vk_out || !vk_len) { set_error("null argument"); return ZKR_ERR_ARG; }
  Generated g;
  int rc = setup_parse_and_run(r1cs_bin, r1cs_len, toxic160, device, g);
  if (!rc) rc = render_websnark(g, pk_out, pk_len);
  if (rc) return rc;
  vk_to_malloc(g, vk_out, vk_len);
  return 0;
}

int zkr_synth_set_shape(unsigned shape) {
  if (shape > 1) { set_error("unknown synthetic circuit shape %u", shape); return ZKR_ERR_ARG; }
  g_shape = shape;
  return 0;
}

int zkr_synth_key(unsigned log_m, unsigned n_public, uint64_t circuit_seed, uint64_t toxic_seed, int device, zkr_key **key_out,
                  void **witness_out, size_t *witness_len, void **aux_out, size_t *aux_len) {
  if (!key_out || !witness_out || !witness_len) { set_error("null argument"); return ZKR_ERR_ARG; }
  Generated g;
  int rc = generate(log_m, n_public, circuit_seed, toxic_seed, device, g);
  if (rc) return rc;
  rc = build_key_from_generated(g, device, key_out);
  if (rc) return rc;
  const Circuit &c = g.circ;
  uint32_t n = c.n;
  std::vector<uint8_t> wb;
  to_std_bytes(c.w, wb);
  *witness_out = malloc(wb.size());
  memcpy(*witness_out, wb.data(), wb.size());
  *witness_len = wb.size();
  if (aux_out && aux_len) {
    std::vector<uint8_t> a, b, cc, tox;
    to_std_bytes(g.sc.a, a); to_std_bytes(g.sc.b, b); to_std_bytes(g.sc.c, cc);
    std::vector<Fr> tv = {g.tox.t, g.tox.alfa, g.tox.beta, g.tox.gamma, g.tox.delta};
    to_std_bytes(tv, tox);
    size_t len = 8 + tox.size() + a.size() * 3 + g.ic_std.size() + 128;
    uint8_t *o = (uint8_t *)malloc(len), *q = o;
    uint64_t n64 = n;
    memcpy(q, &n64, 8); q += 8;
    memcpy(q, tox.data(), tox.size()); q += tox.size();
    memcpy(q, a.data(), a.size()); q += a.size();
    memcpy(q, b.data(), b.size()); q += b.size();
    memcpy(q, cc.data(), cc.size()); q += cc.size();
    memcpy(q, g.ic_std.data(), g.ic_std.size()); q += g.ic_std.size();
    memcpy(q, g.gamma2_std.data(), 128);
    *aux_out = o;
    *aux_len = len;
  }
  return 0;
}

int zkr_synth_witness(unsigned log_m, unsigned n_public, uint64_t circuit_seed, uint64_t witness_seed, void **witness_out, size_t *witness_len) {
  if (!witness_out || !witness_len) { set_error("null argument"); return ZKR_ERR_ARG; }
  if (log_m < 2 || log_m > 26 || n_public + 2 > (1u << log_m)) { set_error("bad synthetic geometry"); return ZKR_ERR_ARG; }
  Circuit c;
  synth_circuit(c, 1u << log_m, n_public, circuit_seed, witness_seed);
  std::vector<uint8_t> wb;
  to_std_bytes(c.w, wb);
  *witness_out = malloc(wb.size());
  memcpy(*witness_out, wb.data(), wb.size());
  *witness_len = wb.size();
  return 0;
}

// verifying key of a synthetic key in the vk_bin layout of zkr_verify: alfa_1, beta_2, delta_2 from the key header
// (Montgomery there), gamma_2 and the IC points from the checker blob of zkr_synth_key
int zkr_synth_vk(const zkr_key *key, const void *aux, size_t aux_len, void **vk_out, size_t *vk_len) {
  if (!key || !aux || !vk_out || !vk_len) { set_error("null argument"); return ZKR_ERR_ARG; }
  const uint8_t *ax = (const uint8_t *)aux;
  uint64_t n64 = 0;
  if (aux_len >= 8) memcpy(&n64, ax, 8);
  const size_t nic = (size_t)key->h.p + 1, ic_off = 8 + 160 + 96 * (size_t)n64;
  if (n64 != key->h.n || aux_len != ic_off + 64 * nic + 128) { set_error("aux blob does not belong to this key"); return ZKR_ERR_ARG; }
  size_t len = 64 + 3 * 128 + 4 + 64 * nic;
  uint8_t *o = (uint8_t *)malloc(len);
  store_g1_std(o, load_g1(key->h.alfa1));
  store_g2_std(o + 64, load_g2(key->h.beta2));
  memcpy(o + 192, ax + ic_off + 64 * nic, 128);
  store_g2_std(o + 320, load_g2(key->h.delta2));
  uint32_t nic32 = (uint32_t)nic;
  memcpy(o + 448, &nic32, 4);
  memcpy(o + 452, ax + ic_off, 64 * nic);
  *vk_out = o;
  *vk_len = len;
  return 0;
}

// the key of a finished setup in the byte layout binarifyProvingKey writes (binarify.ts:143-206)
static int render_websnark(const Generated &g, void **pk_out, size_t *pk_len) {
  const Circuit &c = g.circ;
  uint32_t n = c.n, p = c.p, m = c.m;
  // per-signal columns, constraint index ascending (JS Object.keys order, binarify.ts:104-113)
  std::vector<std::vector<std::pair<uint32_t, Fr>>> colA(n), colB(n);
  for (uint32_t r = 0; r < c.nC; r++) {
    for (uint32_t k = c.rowA[r]; k < c.rowA[r + 1]; k++) colA[c.tA[k].sig].push_back({r, c.tA[k].coef});
    for (uint32_t k = c.rowB[r]; k < c.rowB[r + 1]; k++) colB[c.tB[k].sig].push_back({r, c.tB[k].coef});
  }
  for (uint32_t i = 0; i <= p; i++) colA[i].push_back({c.nC + i, Fr::one()});
  uint64_t size = 40 + 192 + 256;
  for (uint32_t s = 0; s < n; s++) size += 8 + 36ull * (colA[s].size() + colB[s].size());
  size += 64ull * n * 2 + 128ull * n + 64ull * (n - p - 1) + 64ull * m;
  if (size > 0xffffffffull) { set_error("websnark format is limited to 4 GiB (u32 offsets, binarify.ts:155-161)"); return ZKR_ERR_ARG; }
  uint8_t *o = (uint8_t *)malloc(size);
  size_t off = 0;
  auto w32 = [&](uint32_t v) { memcpy(o + off, &v, 4); off += 4; };
  w32(n); w32(p); w32(m);
  size_t ptr_at = off;
  off += 28;
  memcpy(o + off, g.consts, 448);
  off += 448;
  uint32_t ptrs[7];
  auto put_cols = [&](std::vector<std::vector<std::pair<uint32_t, Fr>>> &cols) {
    for (uint32_t s = 0; s < n; s++) {
      w32((uint32_t)cols[s].size());
      for (auto &e : cols[s]) { w32(e.first); memcpy(o + off, e.second.v, 32); off += 32; }
    }
  };
  ptrs[0] = (uint32_t)off; put_cols(colA);
  ptrs[1] = (uint32_t)off; put_cols(colB);
  size_t counts[N_TABLES] = {n, n, n, n - p - 1, m};
  for (int t = 0; t < N_TABLES; t++) {
    ptrs[2 + t] = (uint32_t)off;
    size_t bytes = counts[t] * (t == T_B2 ? 128 : 64);
    if (bytes && hipMemcpy(o + off, g.d_tbl[t], bytes, hipMemcpyDeviceToHost) != hipSuccess) { free(o); set_error("download of key table %d failed", t); return ZKR_ERR_HIP; }
    off += bytes;
  }
  memcpy(o + ptr_at, ptrs, 28);
  if (off != size) { free(o); set_error("internal: websnark size mismatch"); return ZKR_ERR_ARG; }
  *pk_out = o;
  *pk_len = size;
  return 0;
}

int zkr_synth_websnark(unsigned log_m, unsigned n_public, uint64_t circuit_seed, uint64_t toxic_seed, int device, void **pk_out, size_t *pk_len,
                       void **witness_out, size_t *witness_len) {
  if (!pk_out || !pk_len || !witness_out || !witness_len) { set_error("null argument"); return ZKR_ERR_ARG; }
  Generated g;
  int rc = generate(log_m, n_public, circuit_seed, toxic_seed, device, g);
  if (!rc) rc = render_websnark(g, pk_out, pk_len);
  if (rc) return rc;
  std::vector<uint8_t> wb;
  to_std_bytes(g.circ.w, wb);
  *witness_out = malloc(wb.size());
  memcpy(*witness_out, wb.data(), wb.size());
  *witness_len = wb.size();
  return 0;
}

}  // extern "C"
