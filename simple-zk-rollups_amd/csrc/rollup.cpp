// rollup.cpp -- SURVEY 8(f-3): the reference's rollup circuit without circom / snarkjs, host side.
//
//   * witness-side crypto of /root/reference/operator/src/utils/crypto.ts: MiMCSponge-220 `multiHash` (:28-38),
//     `formatPrivKeyForBabyJub` / `genPublicKey` (:58-84), EdDSA-MiMCSponge `sign` / `verify` (:143-177);
//   * a constraint system + witness builder for `BatchProcessTx(batch, depth)`
//     (/root/reference/prover/circuits/batchprocesstx.circom:3-75 over processtx.circom:10-193, eddsa.circom:12-139,
//     merkletree.circom:5-84, hasher.circom:3-30) and for `Withdraw()` (withdraw.circom:4-25): one pass of the gadget program below allocates the signals, emits
//     the rank-1 constraints in the r1cs_bin layout of zkr_setup_r1cs (include/zkr.h) and computes the witness, checking
//     every constraint as it goes -- what `Circuit.calculateWitness` (operator/src/snarks/common.ts:15-17) does for the
//     circom build.  Public signals keep circom's order (output, then the inputs in declaration order; 73 for (2, 6)).
//
// The gadgets are this build's own formulations of the same statements (the constraint count and the private signal
// order differ from circom's output, so keys are made with zkr_setup_r1cs, not taken from a circom build):
// complete twisted-Edwards additions, bit-serial scalar multiplications, one multiplication per path selector.
// One deliberate difference: the reference compares with `GreaterThan(256)` (processtx.circom:86-100), wider than the
// 254-bit field; here amount, fee and balance are range-checked to 250 bits and compared exactly on that range.
//
// Host only (no device code): field arithmetic from field.hpp compiled for the host.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/zkr.h"
#include "field.hpp"

namespace zkr {
void set_error(const char *fmt, ...);
}
using namespace zkr;

namespace zkr { bool rollup_witness_fast_host(uint32_t batch, uint32_t depth, const uint8_t *inputs_std, uint8_t *out_bytes); }  // rollup_gpu.hip

namespace {

// ------------------------------------------------------------------------------------------------ small integers
static Fr fr_u64(uint64_t v) {
  Fr a = Fr::zero();
  a.v[0] = (uint32_t)v;
  a.v[1] = (uint32_t)(v >> 32);
  return to_mont(a);
}
static bool geq_words(const uint32_t *a, const uint32_t *b, int n) {
  for (int i = n - 1; i >= 0; i--)
    if (a[i] != b[i]) return a[i] > b[i];
  return true;
}
static bool fr_read_std(const uint8_t *p, Fr &out) {
  Fr a;
  memcpy(a.v, p, 32);
  if (geq_words(a.v, FrParams::P, 8)) return false;
  out = to_mont(a);
  return true;
}
static void fr_std_words(const Fr &a, uint32_t w[8]) {
  Fr s = from_mont(a);
  memcpy(w, s.v, 32);
}
static void fr_write_std(uint8_t *p, const Fr &a) {
  Fr s = from_mont(a);
  memcpy(p, s.v, 32);
}
// big-endian bytes of any length, reduced mod r (the field operations of the reference reduce their operands)
static Fr fr_from_be(const uint8_t *p, size_t n) {
  const Fr k256 = fr_u64(256);
  Fr acc = Fr::zero();
  for (size_t i = 0; i < n; i++) acc = add(mul(acc, k256), fr_u64(p[i]));
  return acc;
}
// x (nw little-endian words) mod m (8 words), bit-serial
static void mod_words(const uint32_t *x, int nw, const uint32_t m[8], uint32_t out[8]) {
  uint32_t r[9] = {0};
  for (int bit = nw * 32 - 1; bit >= 0; bit--) {
    for (int i = 8; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 31);
    r[0] = (r[0] << 1) | ((x[bit >> 5] >> (bit & 31)) & 1);
    uint32_t m9[9];
    memcpy(m9, m, 32);
    m9[8] = 0;
    if (geq_words(r, m9, 9)) {
      uint64_t br = 0;
      for (int i = 0; i < 9; i++) {
        uint64_t d = (uint64_t)r[i] - m9[i] - br;
        r[i] = (uint32_t)d;
        br = (d >> 63) & 1;
      }
    }
  }
  memcpy(out, r, 32);
}

// ------------------------------------------------------------------------------------------------ keccak-256
static const uint64_t KRC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
                                 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
                                 0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
                                 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                                 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static inline uint64_t rol64(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }
static void keccak_f(uint64_t s[25]) {
  static const int rot[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5y]
  for (int rnd = 0; rnd < 24; rnd++) {
    uint64_t c[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
    for (int x = 0; x < 5; x++) {
      uint64_t d = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
      for (int y = 0; y < 5; y++) s[x + 5 * y] ^= d;
    }
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(s[x + 5 * y], rot[x + 5 * y]);
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    s[0] ^= KRC[rnd];
  }
}
static void keccak256(const uint8_t *data, size_t n, uint8_t out[32]) {
  const size_t rate = 136;
  std::vector<uint8_t> p(data, data + n);
  size_t q = rate - n % rate;
  p.resize(n + q, 0);
  p[n] ^= 0x01;
  p[n + q - 1] ^= 0x80;
  uint64_t s[25] = {0};
  for (size_t off = 0; off < p.size(); off += rate) {
    for (size_t i = 0; i < rate / 8; i++) {
      uint64_t v;
      memcpy(&v, &p[off + 8 * i], 8);
      s[i] ^= v;
    }
    keccak_f(s);
  }
  memcpy(out, s, 32);
}

// ------------------------------------------------------------------------------------------------ MiMCSponge-220
constexpr int NROUNDS = 220;  // hasher.circom:8
struct MimcConstants {
  Fr c[NROUNDS];
  MimcConstants() {
    // circomlib getConstants("mimcsponge", 220) -- the chain the reference deploys on-chain too
    // (contracts/migrations/2_deploy_mimcsponge.js:9-10); pinned by tests/golden/rollup_kat.json
    uint8_t h[32];
    keccak256((const uint8_t *)"mimcsponge", 10, h);
    for (int i = 0; i < NROUNDS; i++) c[i] = Fr::zero();
    for (int i = 1; i < NROUNDS; i++) {
      keccak256(h, 32, h);
      c[i] = fr_from_be(h, 32);
    }
    c[NROUNDS - 1] = Fr::zero();
  }
};
static const MimcConstants &mimc() {
  static const MimcConstants k;
  return k;
}
static inline Fr pow5(const Fr &t) {
  Fr t2 = sqr(t);
  return mul(sqr(t2), t);
}
static void feistel_host(Fr &xl, Fr &xr) {  // key 0 (hasher.circom:21)
  const MimcConstants &k = mimc();
  for (int i = 0; i < NROUNDS; i++) {
    Fr t5 = pow5(add(xl, k.c[i]));
    if (i < NROUNDS - 1) {
      Fr nl = add(xr, t5);
      xr = xl;
      xl = nl;
    } else {
      xr = add(xr, t5);
    }
  }
}
static Fr multihash_host(const Fr *in, size_t n) {  // crypto.ts:28-30
  Fr r = Fr::zero(), c = Fr::zero();
  for (size_t i = 0; i < n; i++) {
    r = add(r, in[i]);
    feistel_host(r, c);
  }
  return r;
}

// ------------------------------------------------------------------------------------------------ BabyJub
// decimal string -> 8 little-endian words (value < 2^256)
static void dec_to_words(const char *s, uint32_t w[8]) {
  memset(w, 0, 32);
  for (; *s; s++) {
    uint64_t carry = (uint64_t)(*s - '0');
    for (int i = 0; i < 8; i++) {
      uint64_t v = (uint64_t)w[i] * 10 + carry;
      w[i] = (uint32_t)v;
      carry = v >> 32;
    }
  }
}
static Fr dec_to_fr(const char *s) {
  Fr a;
  dec_to_words(s, a.v);
  return to_mont(a);
}

struct Bj {
  Fr a, d, b8x, b8y;
  uint32_t suborder[8], suborder_m1[8];
  Bj() {
    a = fr_u64(168700);
    d = fr_u64(168696);
    b8x = dec_to_fr("5299619240641551281634865583518297030282874472190772894086521144482721001553");   // eddsa.circom:88
    b8y = dec_to_fr("16950150798460657717958625567821834550301663161624707787222815936182638968203");  // eddsa.circom:89
    dec_to_words("2736030358979909402780800718157159386076813972158567259200215660948447373040", suborder_m1);  // eddsa.circom:32
    memcpy(suborder, suborder_m1, 32);
    suborder[0] += 1;  // the constant is even
  }
};
static const Bj &bj() {
  static const Bj k;
  return k;
}

struct PtP {  // projective twisted Edwards (X : Y : Z)
  Fr x, y, z;
};
static PtP pt_identity() { return PtP{Fr::zero(), Fr::one(), Fr::one()}; }
static PtP pt_add(const PtP &p, const PtP &q) {  // add-2008-bbjlp, complete on BabyJub
  const Bj &k = bj();
  Fr A = mul(p.z, q.z), B = sqr(A), C = mul(p.x, q.x), D = mul(p.y, q.y);
  Fr E = mul(k.d, mul(C, D)), F = sub(B, E), G = add(B, E);
  Fr x3 = mul(mul(A, F), sub(sub(mul(add(p.x, p.y), add(q.x, q.y)), C), D));
  Fr y3 = mul(mul(A, G), sub(D, mul(k.a, C)));
  return PtP{x3, y3, mul(F, G)};
}
static PtP pt_mul(const PtP &p, const uint32_t *e, int nbits) {
  PtP acc = pt_identity(), q = p;
  for (int i = 0; i < nbits; i++) {
    if ((e[i >> 5] >> (i & 31)) & 1) acc = pt_add(acc, q);
    q = pt_add(q, q);
  }
  return acc;
}
static void pt_affine(const PtP &p, Fr &x, Fr &y) {
  Fr zi = inv(p.z);
  x = mul(p.x, zi);
  y = mul(p.y, zi);
}
static bool pt_on_curve(const Fr &x, const Fr &y) {  // a x^2 + y^2 = 1 + d x^2 y^2
  const Bj &k = bj();
  Fr x2 = sqr(x), y2 = sqr(y);
  return add(mul(k.a, x2), y2) == add(Fr::one(), mul(k.d, mul(x2, y2)));
}

// the hex digits of a field element as TEXT bytes: bigInt2Buffer (crypto.ts:20-22)
static std::string hex_text(const Fr &a) {
  uint32_t w[8];
  fr_std_words(a, w);
  std::string s;
  bool lead = true;
  for (int i = 63; i >= 0; i--) {
    int nib = (w[i >> 3] >> ((i & 7) * 4)) & 15;
    if (lead && nib == 0 && i > 0) continue;
    lead = false;
    s.push_back("0123456789abcdef"[nib]);
  }
  return s;
}
// eddsa.pruneBuffer over the first 32 text bytes; returns the little-endian integer (8 words)
static void pruned_secret(const std::string &h1, uint32_t s[8]) {
  uint8_t b[32] = {0};
  memcpy(b, h1.data(), h1.size() < 32 ? h1.size() : 32);
  b[0] &= 0xF8;
  b[31] &= 0x7F;
  b[31] |= 0x40;
  memcpy(s, b, 32);
}
static void shr3(uint32_t s[8]) {
  for (int i = 0; i < 8; i++) s[i] = (s[i] >> 3) | (i < 7 ? s[i + 1] << 29 : 0);
}
static void pubkey_host(const Fr &priv, Fr &ax, Fr &ay) {  // genPublicKey, crypto.ts:78-84
  uint32_t s[8];
  pruned_secret(hex_text(multihash_host(&priv, 1)), s);
  shr3(s);
  const Bj &k = bj();
  pt_affine(pt_mul(PtP{k.b8x, k.b8y, Fr::one()}, s, 256), ax, ay);
}
static void sign_host(const Fr &priv, const Fr *msg, size_t n, Fr &r8x, Fr &r8y, Fr &S) {  // crypto.ts:143-168
  const Bj &k = bj();
  Fr m = multihash_host(msg, n);
  std::string h1 = hex_text(multihash_host(&priv, 1));
  uint32_t s[8], s3[8];
  pruned_secret(h1, s);
  memcpy(s3, s, 32);
  shr3(s3);
  PtP base{k.b8x, k.b8y, Fr::one()};
  Fr ax, ay;
  pt_affine(pt_mul(base, s3, 256), ax, ay);
  // r = H(h1[32:64] || msgHash as 32 little-endian bytes), the concatenation read as one big-endian integer
  std::vector<uint8_t> cat;
  if (h1.size() > 32) cat.assign(h1.begin() + 32, h1.begin() + (h1.size() < 64 ? h1.size() : 64));
  uint8_t mb[32];
  fr_write_std(mb, m);
  cat.insert(cat.end(), mb, mb + 32);
  Fr pre = fr_from_be(cat.data(), cat.size());
  std::string rb = hex_text(multihash_host(&pre, 1));
  uint32_t rw[16] = {0}, r[8];
  memcpy(rw, rb.data(), rb.size());  // leBuff2int of the text bytes (<= 64)
  mod_words(rw, 16, k.suborder, r);
  pt_affine(pt_mul(base, r, 256), r8x, r8y);
  Fr hin[5] = {r8x, r8y, ax, ay, m};
  uint32_t hm[8];
  fr_std_words(multihash_host(hin, 5), hm);
  // S = (r + hm * s) mod suborder
  uint32_t prod[17] = {0};
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    for (int j = 0; j < 8; j++) {
      uint64_t v = (uint64_t)hm[i] * s[j] + prod[i + j] + c;
      prod[i + j] = (uint32_t)v;
      c = v >> 32;
    }
    prod[i + 8] = (uint32_t)c;
  }
  uint64_t c = 0;
  for (int i = 0; i < 17; i++) {
    uint64_t v = (uint64_t)prod[i] + (i < 8 ? r[i] : 0) + c;
    prod[i] = (uint32_t)v;
    c = v >> 32;
  }
  uint32_t sw[8];
  mod_words(prod, 17, k.suborder, sw);
  Fr sf;
  memcpy(sf.v, sw, 32);
  S = to_mont(sf);
}
// circomlib eddsa.verifyMiMCSponge (crypto.ts:170-177): S*B8 == R8 + hm*(8*A), with curve and range checks
static bool verify_host(const Fr &m, const Fr &r8x, const Fr &r8y, const Fr &S, const Fr &ax, const Fr &ay) {
  const Bj &k = bj();
  if (!pt_on_curve(r8x, r8y) || !pt_on_curve(ax, ay)) return false;
  uint32_t sw[8], hm[8];
  fr_std_words(S, sw);
  if (geq_words(sw, k.suborder, 8)) return false;
  Fr hin[5] = {r8x, r8y, ax, ay, m};
  fr_std_words(multihash_host(hin, 5), hm);
  PtP a8{ax, ay, Fr::one()};
  for (int i = 0; i < 3; i++) a8 = pt_add(a8, a8);
  PtP right = pt_add(PtP{r8x, r8y, Fr::one()}, pt_mul(a8, hm, 256));
  PtP left = pt_mul(PtP{k.b8x, k.b8y, Fr::one()}, sw, 256);
  return mul(left.x, right.z) == mul(right.x, left.z) && mul(left.y, right.z) == mul(right.y, left.z);
}

// ------------------------------------------------------------------------------------------------ constraint builder
struct Term {
  uint32_t sig;
  Fr c;
};
struct LC {
  std::vector<Term> t;
  Fr v;  // value (Montgomery)
};

// Linear combinations carry their terms only while constraints are being serialised; a witness-only pass moves values.
static thread_local bool g_terms = true;

struct Builder {
  bool emit;    // serialise the constraints
  bool check;   // inputs are real: a violated constraint is an error
  std::vector<Fr> w;
  std::vector<uint8_t> out;
  uint32_t n_constraints = 0;
  std::string err;

  Builder(bool emit_, bool check_) : emit(emit_), check(check_) {
    w.push_back(Fr::one());
    g_terms = emit_;
  }

  uint32_t alloc(const Fr &val) {
    w.push_back(val);
    return (uint32_t)w.size() - 1;
  }
  LC sig(uint32_t s) const {
    LC r;
    if (g_terms) r.t.push_back(Term{s, Fr::one()});
    r.v = w[s];
    return r;
  }
  static LC cst(const Fr &k) {
    LC r;
    if (g_terms && !k.is_zero()) r.t.push_back(Term{0, k});
    r.v = k;
    return r;
  }
  static LC zero() { return cst(Fr::zero()); }
  static LC one() { return cst(Fr::one()); }

  void put_lc(const LC &a) {
    // merge repeated signals, drop zero coefficients
    std::vector<Term> m;
    m.reserve(a.t.size());
    for (const Term &x : a.t) {
      bool found = false;
      for (Term &y : m)
        if (y.sig == x.sig) {
          y.c = add(y.c, x.c);
          found = true;
          break;
        }
      if (!found) m.push_back(x);
    }
    uint32_t k = 0;
    for (const Term &y : m) k += !y.c.is_zero();
    size_t o = out.size();
    out.resize(o + 4 + 36 * (size_t)k);
    memcpy(&out[o], &k, 4);
    o += 4;
    for (const Term &y : m) {
      if (y.c.is_zero()) continue;
      memcpy(&out[o], &y.sig, 4);
      fr_write_std(&out[o + 4], y.c);
      o += 36;
    }
  }
  void enforce(const LC &a, const LC &b, const LC &c, const char *what) {
    if (check && err.empty() && !(mul(a.v, b.v) == c.v)) {
      err = what;
    }
    if (emit) {
      put_lc(a);
      put_lc(b);
      put_lc(c);
    }
    n_constraints++;
  }
  // new signal = a * b
  // a constraint whose right-hand side was just computed from its left-hand side: nothing to check
  void define(const LC &a, const LC &b, const LC &c) {
    if (emit) {
      put_lc(a);
      put_lc(b);
      put_lc(c);
    }
    n_constraints++;
  }
  LC mul_sig(const LC &a, const LC &b, const char *) {
    LC s = sig(alloc(mul(a.v, b.v)));
    define(a, b, s);
    return s;
  }
};

static LC operator+(const LC &a, const LC &b) {
  LC r;
  if (g_terms) {
    r.t.reserve(a.t.size() + b.t.size());
    r.t = a.t;
    r.t.insert(r.t.end(), b.t.begin(), b.t.end());
  }
  r.v = add(a.v, b.v);
  return r;
}
static LC operator*(const LC &a, const Fr &k) {
  LC r;
  if (g_terms) {
    r.t.reserve(a.t.size());
    for (const Term &x : a.t) r.t.push_back(Term{x.sig, mul(x.c, k)});
  }
  r.v = mul(a.v, k);
  return r;
}
static LC operator-(const LC &a, const LC &b) {
  LC r;
  if (g_terms) {
    r.t.reserve(a.t.size() + b.t.size());
    r.t = a.t;
    for (const Term &x : b.t) r.t.push_back(Term{x.sig, neg(x.c)});
  }
  r.v = sub(a.v, b.v);
  return r;
}

// ---- MiMCSponge gadget (circomlib MiMCFeistel: t2 = t*t, t4 = t2*t2, x' = x_r + t4*t; 3 constraints per round)
static void feistel_gadget(Builder &B, LC &xl, LC &xr) {
  const MimcConstants &k = mimc();
  for (int i = 0; i < NROUNDS; i++) {
    LC t = k.c[i].is_zero() ? xl : xl + Builder::cst(k.c[i]);
    LC t2 = B.mul_sig(t, t, "MiMC t^2");
    LC t4 = B.mul_sig(t2, t2, "MiMC t^4");
    LC nx = B.sig(B.alloc(add(xr.v, mul(t4.v, t.v))));
    B.define(t4, t, nx - xr);
    if (i < NROUNDS - 1) {
      xr = xl;
      xl = nx;
    } else {
      xr = nx;
    }
  }
}
static LC multihash_gadget(Builder &B, const std::vector<LC> &in) {  // Hasher(length), key 0 (hasher.circom:3-16)
  LC r = Builder::zero(), c = Builder::zero();
  for (const LC &x : in) {
    r = r + x;
    feistel_gadget(B, r, c);
  }
  return r;
}

// ---- bits
static std::vector<LC> num2bits(Builder &B, const LC &in, int n, const char *what) {  // circomlib Num2Bits(n)
  uint32_t w[8];
  fr_std_words(in.v, w);
  std::vector<LC> bits;
  LC sum = Builder::zero();
  Fr p2 = Fr::one();
  for (int i = 0; i < n; i++) {
    uint32_t bit = i < 256 ? (w[i >> 5] >> (i & 31)) & 1 : 0;
    LC b = B.sig(B.alloc(bit ? Fr::one() : Fr::zero()));
    B.enforce(b, b - Builder::one(), Builder::zero(), "bit is 0 or 1");
    sum = sum + b * p2;
    p2 = dbl(p2);
    bits.push_back(b);
  }
  B.enforce(sum, Builder::one(), in, what);
  return bits;
}
// 1 if the number with these bits is greater than the constant c (n bits), else 0; one constraint per bit
static LC bits_gt_const(Builder &B, const std::vector<LC> &bits, const uint32_t *c) {
  LC gt = Builder::zero();
  bool is_const_zero = true;
  for (size_t i = 0; i < bits.size(); i++) {
    uint32_t cb = (c[i >> 5] >> (i & 31)) & 1;
    if (cb == 0) {
      if (is_const_zero) {
        gt = bits[i];
        is_const_zero = false;
      } else {
        gt = bits[i] + gt - B.mul_sig(bits[i], gt, "compare bit (or)");
      }
    } else if (!is_const_zero) {
      gt = B.mul_sig(bits[i], gt, "compare bit (and)");
    }
  }
  return gt;
}

// ---- BabyJub gadgets
struct PtL {
  LC x, y;
};
// 1/z for every element (0 for a zero element), one field inversion in all
static void batch_inverse(std::vector<Fr> &z) {
  std::vector<Fr> pre(z.size());
  Fr acc = Fr::one();
  for (size_t i = 0; i < z.size(); i++) {
    pre[i] = acc;
    if (!z[i].is_zero()) acc = mul(acc, z[i]);
  }
  Fr ia = inv(acc);
  for (size_t i = z.size(); i-- > 0;) {
    if (z[i].is_zero()) continue;
    Fr zi = mul(ia, pre[i]);
    ia = mul(ia, z[i]);
    z[i] = zi;
  }
}
// affine coordinates of a list of projective points (x0, y0, x1, y1, ...)
static std::vector<Fr> affine_all(const std::vector<PtP> &pts) {
  std::vector<Fr> z(pts.size());
  for (size_t i = 0; i < pts.size(); i++) z[i] = pts[i].z;
  batch_inverse(z);
  std::vector<Fr> out(2 * pts.size());
  for (size_t i = 0; i < pts.size(); i++) out[2 * i] = mul(pts[i].x, z[i]), out[2 * i + 1] = mul(pts[i].y, z[i]);
  return out;
}

// the complete addition law, 6 constraints; `known` = the sum's affine coordinates when a caller has them already
static PtL edwards_add(Builder &B, const PtL &p, const PtL &q, const Fr *known = nullptr) {
  const Bj &k = bj();
  LC beta = B.mul_sig(p.x, q.y, "edwards beta");
  LC gamma = B.mul_sig(p.y, q.x, "edwards gamma");
  LC delta = B.mul_sig(p.y - p.x * k.a, q.x + q.y, "edwards delta");
  LC tau = B.mul_sig(beta, gamma, "edwards tau");
  LC dt = tau * k.d;
  LC dx = Builder::one() + dt, dy = Builder::one() - dt;
  LC nx = beta + gamma, ny = delta + beta * k.a - gamma;
  Fr xv, yv;
  if (known) {
    xv = known[0], yv = known[1];
  } else {
    Fr di = inv(mul(dx.v, dy.v));
    xv = mul(nx.v, mul(di, dy.v)), yv = mul(ny.v, mul(di, dx.v));
  }
  LC xo = B.sig(B.alloc(xv));
  LC yo = B.sig(B.alloc(yv));
  B.enforce(dx, xo, nx, "edwards x");
  B.enforce(dy, yo, ny, "edwards y");
  return PtL{xo, yo};
}
// sum_i e_i 2^i P for a variable point: per bit a selected addition (2 + 6) and a doubling (6)
static PtL scalar_mul_any(Builder &B, const std::vector<LC> &e, PtL q) {
  // values first, in projective coordinates, so that the whole chain costs one field inversion
  const size_t n = e.size();
  std::vector<PtP> pts;  // acc_1, q_1, acc_2, q_2, ...
  pts.reserve(2 * n);
  {
    PtP qq{q.x.v, q.y.v, Fr::one()}, acc = e[0].v.is_zero() ? pt_identity() : qq;
    for (size_t i = 1; i < n; i++) {
      qq = pt_add(qq, qq);
      if (!e[i].v.is_zero()) acc = pt_add(acc, qq);
      pts.push_back(acc);
      pts.push_back(qq);
    }
  }
  std::vector<Fr> aff = affine_all(pts);
  PtL acc;
  for (size_t i = 0; i < n; i++) {
    PtL sel{B.mul_sig(e[i], q.x, "select x"), B.mul_sig(e[i], q.y - Builder::one(), "select y") + Builder::one()};
    acc = i == 0 ? sel : edwards_add(B, acc, sel, &aff[4 * (i - 1)]);
    if (i + 1 < n) q = edwards_add(B, q, q, &aff[4 * i + 2]);
  }
  return acc;
}
// sum_i e_i 2^i BASE8: the multiples are constants, the selection is linear, one addition per bit
struct Base8Table {
  std::vector<Fr> x, y;
  Base8Table() {
    const Bj &k = bj();
    PtP q{k.b8x, k.b8y, Fr::one()};
    for (int i = 0; i < 253; i++) {
      Fr ax, ay;
      pt_affine(q, ax, ay);
      x.push_back(ax);
      y.push_back(ay);
      q = pt_add(q, q);
    }
  }
};
static PtL scalar_mul_base8(Builder &B, const std::vector<LC> &e) {
  static const Base8Table tab;
  std::vector<PtP> pts;
  pts.reserve(e.size());
  {
    PtP acc = e[0].v.is_zero() ? pt_identity() : PtP{tab.x[0], tab.y[0], Fr::one()};
    for (size_t i = 1; i < e.size(); i++) {
      if (!e[i].v.is_zero()) acc = pt_add(acc, PtP{tab.x[i], tab.y[i], Fr::one()});
      pts.push_back(acc);
    }
  }
  std::vector<Fr> aff = affine_all(pts);
  PtL acc;
  for (size_t i = 0; i < e.size(); i++) {
    PtL sel{e[i] * tab.x[i], e[i] * sub(tab.y[i], Fr::one()) + Builder::one()};
    acc = i == 0 ? sel : edwards_add(B, acc, sel, &aff[2 * (i - 1)]);
  }
  return acc;
}
static LC is_zero(Builder &B, const LC &in) {  // circomlib IsZero
  LC iv = B.sig(B.alloc(in.v.is_zero() ? Fr::zero() : inv(in.v)));
  LC out = B.sig(B.alloc(in.v.is_zero() ? Fr::one() : Fr::zero()));
  B.enforce(in, iv, Builder::one() - out, "IsZero inverse");
  B.enforce(in, out, Builder::zero(), "IsZero product");
  return out;
}

// ---- EdDSAMiMCSpongeVerifierPatched (eddsa.circom:12-110): returns `valid`
static LC eddsa_verify_gadget(Builder &B, const LC &ax, const LC &ay, const LC &S, const LC &r8x, const LC &r8y, const LC &M) {
  const Bj &k = bj();
  std::vector<LC> sbits = num2bits(B, S, 253, "S fits 253 bits");
  B.enforce(bits_gt_const(B, sbits, k.suborder_m1), Builder::one(), Builder::zero(), "S below the subgroup order");  // eddsa.circom:28-38
  LC h = multihash_gadget(B, {r8x, r8y, ax, ay, M});                                                                   // :41-47
  std::vector<LC> hbits = num2bits(B, h, 254, "hash bits");
  uint32_t rm1[8];
  memcpy(rm1, FrParams::P, 32);
  rm1[0] -= 1;
  B.enforce(bits_gt_const(B, hbits, rm1), Builder::one(), Builder::zero(), "hash bits are the canonical ones");  // Num2Bits_strict, :49-50
  PtL a2 = edwards_add(B, PtL{ax, ay}, PtL{ax, ay});                                                                 // :56-64
  PtL a4 = edwards_add(B, a2, a2);
  PtL a8 = edwards_add(B, a4, a4);
  B.enforce(is_zero(B, a4.x), Builder::one(), Builder::zero(), "A is not a small-order point");  // :67-69
  PtL right2 = scalar_mul_any(B, hbits, a8);                                                      // :71-76
  PtL right = edwards_add(B, PtL{r8x, r8y}, right2);                                              // :80-84
  PtL left = scalar_mul_base8(B, sbits);                                                          // :87-94
  LC ex = is_zero(B, left.x - right.x), ey = is_zero(B, left.y - right.y);                        // :97-103
  return is_zero(B, ex + ey - Builder::cst(fr_u64(2)));                                           // :105-109
}

// ---- MerkleTreeRootConstructor (merkletree.circom:33-63); path bits are boolean by construction (Num2Bits)
static LC merkle_root_gadget(Builder &B, LC cur, const std::vector<LC> &path, const std::vector<LC> &idx) {
  for (size_t i = 0; i < path.size(); i++) {
    // index 0: (left, right) = (cur, sibling); index 1: (sibling, cur)   (merkletree.circom:3-31)
    LC left = cur + B.mul_sig(idx[i], path[i] - cur, "path selector");
    LC right = path[i] + cur - left;
    cur = multihash_gadget(B, {left, right});
  }
  return cur;
}

struct TxIn {  // the inputs of one ProcessTx (processtx.circom:13-67)
  LC root, tx[8], spk[2], sbal, snonce, rpk[2], rbal, rnonce, iroot;
  std::vector<LC> spath, rpath, ipath;
};

// a > b for a, b already known to be below 2^250
static LC greater_than_250(Builder &B, const LC &a, const LC &b, const char *what) {
  Fr p251 = Fr::one();
  for (int i = 0; i < 251; i++) p251 = dbl(p251);
  std::vector<LC> d = num2bits(B, a - b - Builder::one() + Builder::cst(p251), 252, what);
  return d[251];
}

static LC process_tx_gadget(Builder &B, const TxIn &in, uint32_t depth) {
  std::vector<LC> sidx = num2bits(B, in.tx[0], depth, "sender index fits the tree");      // processtx.circom:51-52
  std::vector<LC> ridx = num2bits(B, in.tx[1], depth, "recipient index fits the tree");   // :59-60 (and :68-69, the same bits)
  // 1.1 signature over (from, to, amount, fee, nonce)   (:72-82, eddsa.circom:113-139)
  LC msg = multihash_gadget(B, {in.tx[0], in.tx[1], in.tx[2], in.tx[3], in.tx[4]});
  LC valid = eddsa_verify_gadget(B, in.spk[0], in.spk[1], in.tx[7], in.tx[5], in.tx[6], msg);
  B.enforce(valid, Builder::one(), Builder::one(), "transaction signature is valid");
  // 1.2 nonce, amount, fee   (:85-95)
  B.enforce(in.tx[4], Builder::one(), in.snonce + Builder::one(), "nonce is the sender's nonce + 1");
  num2bits(B, in.tx[2], 250, "amount fits 250 bits");
  num2bits(B, in.tx[3], 250, "fee fits 250 bits");
  num2bits(B, in.sbal, 250, "sender balance fits 250 bits");
  B.enforce(is_zero(B, in.tx[2]), Builder::one(), Builder::zero(), "amount > 0");
  B.enforce(is_zero(B, in.tx[3]), Builder::one(), Builder::zero(), "fee > 0");
  // 2. balance > amount + fee   (:98-101)
  B.enforce(greater_than_250(B, in.sbal, in.tx[2] + in.tx[3], "balance comparison"), Builder::one(), Builder::one(), "sender balance > amount + fee");
  // 3. both leaves are in the tree   (:106-136)
  LC sleaf = multihash_gadget(B, {in.spk[0], in.spk[1], in.sbal, in.snonce});
  LC rleaf = multihash_gadget(B, {in.rpk[0], in.rpk[1], in.rbal, in.rnonce});
  B.enforce(merkle_root_gadget(B, sleaf, in.spath, sidx), Builder::one(), in.root, "sender leaf is in the balance tree");
  B.enforce(merkle_root_gadget(B, rleaf, in.rpath, ridx), Builder::one(), in.root, "recipient leaf is in the balance tree");
  // 4. new leaves   (:139-173)
  LC nsbal = in.sbal - in.tx[2] - in.tx[3];
  LC nsleaf = multihash_gadget(B, {in.spk[0], in.spk[1], nsbal, in.tx[4]});
  LC same = is_zero(B, in.tx[0] - in.tx[1]);
  LC selbal = in.rbal + B.mul_sig(same, nsbal - in.rbal, "recipient balance selector");
  LC selnonce = in.rnonce + B.mul_sig(same, in.tx[4] - in.rnonce, "recipient nonce selector");
  LC nrleaf = multihash_gadget(B, {in.rpk[0], in.rpk[1], selbal + in.tx[2], selnonce});
  // 5. the two updates   (:176-192)
  B.enforce(merkle_root_gadget(B, nsleaf, in.spath, sidx), Builder::one(), in.iroot, "intermediate root after the sender update");
  return merkle_root_gadget(B, nrleaf, in.ipath, ridx);
}

constexpr uint32_t PER_TX = 8 + 2 + 1 + 1 + 2 + 1 + 1 + 1 + 1;  // scalars + fixed arrays per tx, without the 3 paths

static uint32_t n_public_of(uint32_t batch, uint32_t depth) { return 1 + batch * (PER_TX + 3 * depth); }

struct Layout {  // circom's order: the output, then every input array in declaration order, each flattened over the batch
  uint32_t o_root, o_tx, o_spk, o_sbal, o_snonce, o_spath, o_rpk, o_rbal, o_rnonce, o_rpath, o_iroot, o_ipath;
  Layout(uint32_t batch, uint32_t depth) {
    uint32_t at = 2;
    auto take = [&](uint32_t count) {
      uint32_t first = at;
      at += batch * count;
      return first;
    };
    o_root = take(1), o_tx = take(8), o_spk = take(2), o_sbal = take(1), o_snonce = take(1), o_spath = take(depth), o_rpk = take(2);
    o_rbal = take(1), o_rnonce = take(1), o_rpath = take(depth), o_iroot = take(1), o_ipath = take(depth);
  }
};
static TxIn tx_inputs(const Builder &B, const Layout &L, uint32_t i, uint32_t depth) {
  TxIn in;
  in.root = B.sig(L.o_root + i);
  for (int j = 0; j < 8; j++) in.tx[j] = B.sig(L.o_tx + 8 * i + j);
  for (int j = 0; j < 2; j++) in.spk[j] = B.sig(L.o_spk + 2 * i + j), in.rpk[j] = B.sig(L.o_rpk + 2 * i + j);
  in.sbal = B.sig(L.o_sbal + i), in.snonce = B.sig(L.o_snonce + i), in.rbal = B.sig(L.o_rbal + i), in.rnonce = B.sig(L.o_rnonce + i);
  in.iroot = B.sig(L.o_iroot + i);
  for (uint32_t j = 0; j < depth; j++) {
    in.spath.push_back(B.sig(L.o_spath + depth * i + j));
    in.rpath.push_back(B.sig(L.o_rpath + depth * i + j));
    in.ipath.push_back(B.sig(L.o_ipath + depth * i + j));
  }
  return in;
}

// BatchProcessTx(batch, depth) (batchprocesstx.circom:3-75), structure pass: signals and constraints in order.
static void batch_gadget(Builder &B, uint32_t batch, uint32_t depth) {
  const uint32_t p = n_public_of(batch, depth);
  for (uint32_t i = 1; i <= p; i++) B.alloc(Fr::zero());
  const Layout L(batch, depth);
  LC prev;
  for (uint32_t i = 0; i < batch; i++) {
    TxIn in = tx_inputs(B, L, i, depth);
    if (i > 0) B.enforce(prev, Builder::one(), in.root, "a transaction starts from the root the previous one produced");  // :67-69
    prev = process_tx_gadget(B, in, depth);
  }
  B.w[1] = prev.v;  // newBalanceTreeRoot (:72)
  B.enforce(prev, Builder::one(), B.sig(1), "newBalanceTreeRoot");
}

// private signals of one ProcessTx(depth) (the structure does not depend on the inputs); cached per depth
static uint32_t tx_private_count(uint32_t depth) {
  static std::mutex mu;
  static uint32_t cache[33] = {0};
  std::lock_guard<std::mutex> lk(mu);
  if (!cache[depth]) {
    Builder B(false, false);
    const uint32_t p = n_public_of(1, depth);
    for (uint32_t i = 1; i <= p; i++) B.alloc(Fr::zero());
    process_tx_gadget(B, tx_inputs(B, Layout(1, depth), 0, depth), depth);
    cache[depth] = (uint32_t)B.w.size() - p - 1;
  }
  return cache[depth];
}

static void batch_gadget(Builder &B, uint32_t batch, uint32_t depth);
// ZKR_WITNESS_VERIFY=1 (CI / debugging): index of the first constraint of the EMITTED system (zkr_rollup_r1cs, cached per
// geometry) that the witness violates, -1 when it satisfies them all.  The fast path below returns the value program's witness
// whenever its own statement checks pass; this is the net under that hand-written list -- a missed statement would otherwise
// only show later, as an invalid proof.
static long first_violated_constraint(uint32_t batch, uint32_t depth, const uint8_t *witness_std, size_t n_signals) {
  static std::mutex mu;
  static std::map<std::pair<uint32_t, uint32_t>, std::vector<uint8_t>> systems;
  const std::vector<uint8_t> *sys = nullptr;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = systems.find({batch, depth});
    if (it == systems.end()) {
      Builder B(true, false);
      B.out.resize(12);
      batch_gadget(B, batch, depth);
      uint32_t hdr[3] = {(uint32_t)B.w.size(), n_public_of(batch, depth), B.n_constraints};
      memcpy(B.out.data(), hdr, 12);
      it = systems.emplace(std::make_pair(batch, depth), std::move(B.out)).first;
    }
    sys = &it->second;
  }
  uint32_t hdr[3];
  memcpy(hdr, sys->data(), 12);
  if (hdr[0] != n_signals) return 0;
  std::vector<Fr> w(n_signals);
  for (size_t i = 0; i < n_signals; i++)
    if (!fr_read_std(witness_std + 32 * i, w[i])) return 0;
  size_t off = 12;
  for (uint32_t c = 0; c < hdr[2]; c++) {
    Fr v[3];
    for (int side = 0; side < 3; side++) {
      uint32_t k;
      memcpy(&k, sys->data() + off, 4);
      off += 4;
      Fr acc = Fr::zero();
      for (uint32_t e = 0; e < k; e++, off += 36) {
        uint32_t sig;
        memcpy(&sig, sys->data() + off, 4);
        Fr cf;
        if (sig >= n_signals || !fr_read_std(sys->data() + off + 4, cf)) return (long)c;
        acc = add(acc, mul(cf, w[sig]));
      }
      v[side] = acc;
    }
    if (!(mul(v[0], v[1]) == v[2])) return (long)c;
  }
  return -1;
}

// Witness pass: the transactions of a batch only meet in the root chain, so each one is built on its own thread
// (same gadget code, same signal order) and writes its private signals straight into its slice of the result, laid out
// as binarifyWitness does (32 B standard form per signal; malloc'ed, the caller frees).
static bool batch_witness(uint32_t batch, uint32_t depth, const Fr *inputs, const uint8_t *inputs_std, uint8_t **out_buf, size_t *out_len, std::string &err) {
  const uint32_t p = n_public_of(batch, depth), K = tx_private_count(depth);
  const Layout L(batch, depth);
  std::vector<Fr> pub(p + 1, Fr::zero());
  pub[0] = Fr::one();
  for (uint32_t i = 2; i <= p; i++) pub[i] = inputs[i - 2];
  const size_t total = 32 * ((size_t)p + 1 + (size_t)batch * K);
  uint8_t *out = (uint8_t *)malloc(total);
  if (!out) { err = "out of memory"; return false; }
  // Fast path: the value program the GPU builder runs (rollup_witness.hpp), on host threads -- a task per (transaction, part)
  // instead of a thread per transaction, no linear-combination bookkeeping: 5.9 -> ~2 ms for a (2, 6) batch.  tests/test_rollup.py
  // holds the two builders equal signal for signal; whenever a statement fails the gadget builder below runs and names it.
  const bool no_fast = getenv("ZKR_WITNESS_GADGETS") != nullptr;  // read per call: the tests switch builders inside one process
  if (!no_fast && zkr::rollup_witness_fast_host(batch, depth, inputs_std, out)) {
    if (const char *e = getenv("ZKR_WITNESS_VERIFY"); e && atoi(e) != 0) {
      const long bad = first_violated_constraint(batch, depth, out, total / 32);
      if (bad >= 0) {
        free(out);
        err = "internal: the fast witness builder's result violates constraint " + std::to_string(bad) + " of the emitted system (ZKR_WITNESS_VERIFY)";
        return false;
      }
    }
    *out_buf = out;
    *out_len = total;
    return true;
  }
  std::vector<Fr> roots(batch);
  std::vector<std::string> errs(batch);
  auto one_tx = [&](uint32_t i) {
    Builder B(false, true);
    B.w.reserve(p + 1 + K);
    B.w = pub;
    roots[i] = process_tx_gadget(B, tx_inputs(B, L, i, depth), depth).v;
    errs[i] = B.err;
    if (B.w.size() != (size_t)p + 1 + K) { errs[i] = "internal: private signal count differs from the structure pass"; return; }
    uint8_t *dst = out + 32 * ((size_t)p + 1 + (size_t)i * K);
    for (size_t k = 0; k < K; k++) fr_write_std(dst + 32 * k, B.w[p + 1 + k]);
  };
  unsigned hw = std::thread::hardware_concurrency();
  const uint32_t nthreads = std::min<uint32_t>(batch, hw ? hw : 1);
  if (nthreads <= 1) {
    for (uint32_t i = 0; i < batch; i++) one_tx(i);
  } else {
    std::vector<std::thread> th;
    std::atomic<uint32_t> next{0};
    for (uint32_t t = 0; t < nthreads; t++)
      th.emplace_back([&] {
        for (uint32_t i; (i = next.fetch_add(1)) < batch;) one_tx(i);
      });
    for (auto &t : th) t.join();
  }
  for (uint32_t i = 0; i < batch; i++) {  // first violated statement in circuit order
    if (i > 0 && !(roots[i - 1] == pub[L.o_root + i])) err = "transaction " + std::to_string(i) + " violates: a transaction starts from the root the previous one produced";
    else if (!errs[i].empty()) err = "transaction " + std::to_string(i) + " violates: " + errs[i];
    if (!err.empty()) { free(out); return false; }
  }
  pub[1] = roots[batch - 1];  // newBalanceTreeRoot (:72)
  for (uint32_t k = 0; k <= p; k++) fr_write_std(out + 32 * k, pub[k]);
  *out_buf = out;
  *out_len = total;
  return true;
}

// Withdraw() (withdraw.circom:4-25 over publickeyderivation.circom:5-27): signals 1, 2 = publicKey (outputs),
// 3 = nullifier (public input), 4 = privateKey (private input: the formatted key, crypto.ts:58-76).
static void withdraw_gadget(Builder &B, const Fr &priv, const Fr &nullifier) {
  B.alloc(Fr::zero());
  B.alloc(Fr::zero());
  LC nul = B.sig(B.alloc(nullifier)), key = B.sig(B.alloc(priv));
  std::vector<LC> bits = num2bits(B, key, 253, "private key fits 253 bits");  // publickeyderivation.circom:12-13
  PtL pk = scalar_mul_base8(B, bits);                                          // :20-23
  multihash_gadget(B, {pk.x, pk.y, nul});                                      // withdraw.circom:15-19 (binds the nullifier)
  B.w[1] = pk.x.v, B.w[2] = pk.y.v;
  B.enforce(pk.x, Builder::one(), B.sig(1), "publicKey[0]");                   // :21-22
  B.enforce(pk.y, Builder::one(), B.sig(2), "publicKey[1]");
}

static int check_geometry(uint32_t batch, uint32_t depth) {
  if (batch < 1 || batch > 4096 || depth < 1 || depth > 32) {
    set_error("rollup circuit: batch %u / depth %u out of range (1..4096, 1..32)", batch, depth);
    return ZKR_ERR_ARG;
  }
  return ZKR_OK;
}

static void *dup_bytes(const void *p, size_t n) {
  void *o = malloc(n ? n : 1);
  if (o && n) memcpy(o, p, n);
  return o;
}

}  // namespace

namespace zkr {
// the 220 round constants (Montgomery form) for the device kernels of rollup_gpu.hip
const Fr *mimc_round_constants() { return mimc().c; }
// what the device witness builder (rollup_gpu.hip) shares with the gadget program above: curve constants, the fixed-base
// table of scalar_mul_base8, and the geometry of the signal vector
void rollup_device_constants(Fr *a, Fr *d, uint32_t suborder_m1[8], Fr *b8x253, Fr *b8y253) {
  const Bj &k = bj();
  *a = k.a;
  *d = k.d;
  memcpy(suborder_m1, k.suborder_m1, 32);
  static const Base8Table tab;
  for (int i = 0; i < 253; i++) b8x253[i] = tab.x[i], b8y253[i] = tab.y[i];
}
uint32_t rollup_tx_private_count(uint32_t depth) { return tx_private_count(depth); }
uint32_t rollup_n_public(uint32_t batch, uint32_t depth) { return n_public_of(batch, depth); }
int rollup_check_geometry(uint32_t batch, uint32_t depth) { return check_geometry(batch, depth); }
}  // namespace zkr

extern "C" {

int zkr_mimcsponge_multihash(const uint8_t *in, size_t n, uint8_t out[32]) {
  if ((!in && n) || !out) { set_error("null argument"); return ZKR_ERR_ARG; }
  std::vector<Fr> v(n);
  for (size_t i = 0; i < n; i++) {  // operands are taken mod r, as the reference's field operations do
    uint8_t be[32];
    for (int j = 0; j < 32; j++) be[j] = in[32 * i + 31 - j];
    v[i] = fr_from_be(be, 32);
  }
  fr_write_std(out, multihash_host(v.data(), n));
  return ZKR_OK;
}

int zkr_babyjub_pubkey(const uint8_t priv[32], uint8_t pub[64]) {
  Fr k, x, y;
  if (!priv || !pub) { set_error("null argument"); return ZKR_ERR_ARG; }
  if (!fr_read_std(priv, k)) { set_error("private key >= r (crypto.ts:79)"); return ZKR_ERR_ARG; }
  pubkey_host(k, x, y);
  fr_write_std(pub, x);
  fr_write_std(pub + 32, y);
  return ZKR_OK;
}

int zkr_eddsa_sign(const uint8_t priv[32], const uint8_t *msg, size_t n, uint8_t sig[96]) {
  Fr k;
  if (!priv || (!msg && n) || !sig) { set_error("null argument"); return ZKR_ERR_ARG; }
  if (!fr_read_std(priv, k)) { set_error("private key >= r"); return ZKR_ERR_ARG; }
  std::vector<Fr> m(n);
  for (size_t i = 0; i < n; i++)
    if (!fr_read_std(msg + 32 * i, m[i])) { set_error("message element %zu >= r", i); return ZKR_ERR_ARG; }
  Fr r8x, r8y, S;
  sign_host(k, m.data(), n, r8x, r8y, S);
  fr_write_std(sig, r8x);
  fr_write_std(sig + 32, r8y);
  fr_write_std(sig + 64, S);
  return ZKR_OK;
}

int zkr_eddsa_verify(const uint8_t *msg, size_t n, const uint8_t sig[96], const uint8_t pub[64], int *valid) {
  if ((!msg && n) || !sig || !pub || !valid) { set_error("null argument"); return ZKR_ERR_ARG; }
  std::vector<Fr> m(n);
  for (size_t i = 0; i < n; i++)
    if (!fr_read_std(msg + 32 * i, m[i])) { set_error("message element %zu >= r", i); return ZKR_ERR_ARG; }
  Fr s[3], a[2];
  *valid = 0;
  for (int i = 0; i < 3; i++)
    if (!fr_read_std(sig + 32 * i, s[i])) return ZKR_OK;  // not a field element: not a valid signature
  for (int i = 0; i < 2; i++)
    if (!fr_read_std(pub + 32 * i, a[i])) return ZKR_OK;
  *valid = verify_host(multihash_host(m.data(), n), s[0], s[1], s[2], a[0], a[1]) ? 1 : 0;
  return ZKR_OK;
}

int zkr_babyjub_format_privkey(const uint8_t priv[32], uint8_t out[32]) {
  Fr k;
  if (!priv || !out) { set_error("null argument"); return ZKR_ERR_ARG; }
  if (!fr_read_std(priv, k)) { set_error("private key >= r"); return ZKR_ERR_ARG; }
  uint32_t s[8];
  pruned_secret(hex_text(multihash_host(&k, 1)), s);
  shr3(s);
  memcpy(out, s, 32);
  return ZKR_OK;
}

int zkr_withdraw_r1cs(void **r1cs_bin, size_t *r1cs_len) {
  if (!r1cs_bin || !r1cs_len) { set_error("null argument"); return ZKR_ERR_ARG; }
  Builder B(true, false);
  B.out.resize(12);
  withdraw_gadget(B, Fr::zero(), Fr::zero());
  uint32_t hdr[3] = {(uint32_t)B.w.size(), 3, B.n_constraints};
  memcpy(B.out.data(), hdr, 12);
  *r1cs_bin = dup_bytes(B.out.data(), B.out.size());
  *r1cs_len = B.out.size();
  if (!*r1cs_bin) { set_error("out of memory"); return ZKR_ERR_ARG; }
  return ZKR_OK;
}

int zkr_withdraw_witness(const uint8_t private_key[32], const uint8_t nullifier[32], void **witness_bin, size_t *witness_len) {
  if (!private_key || !nullifier || !witness_bin || !witness_len) { set_error("null argument"); return ZKR_ERR_ARG; }
  Fr k, nul;
  if (!fr_read_std(private_key, k) || !fr_read_std(nullifier, nul)) { set_error("input >= r"); return ZKR_ERR_ARG; }
  Builder B(false, true);
  withdraw_gadget(B, k, nul);
  if (!B.err.empty()) { set_error("withdraw circuit violates: %s", B.err.c_str()); return ZKR_ERR_UNSATISFIED; }
  std::vector<uint8_t> out(32 * B.w.size());
  for (size_t i = 0; i < B.w.size(); i++) fr_write_std(&out[32 * i], B.w[i]);
  *witness_bin = dup_bytes(out.data(), out.size());
  *witness_len = out.size();
  if (!*witness_bin) { set_error("out of memory"); return ZKR_ERR_ARG; }
  return ZKR_OK;
}

int zkr_rollup_info(uint32_t batch, uint32_t depth, uint32_t *n_vars, uint32_t *n_public, uint32_t *n_constraints) {
  int rc = check_geometry(batch, depth);
  if (rc) return rc;
  Builder B(false, false);
  batch_gadget(B, batch, depth);
  if (n_vars) *n_vars = (uint32_t)B.w.size();
  if (n_public) *n_public = n_public_of(batch, depth);
  if (n_constraints) *n_constraints = B.n_constraints;
  return ZKR_OK;
}

int zkr_rollup_r1cs(uint32_t batch, uint32_t depth, void **r1cs_bin, size_t *r1cs_len) {
  if (!r1cs_bin || !r1cs_len) { set_error("null argument"); return ZKR_ERR_ARG; }
  int rc = check_geometry(batch, depth);
  if (rc) return rc;
  Builder B(true, false);
  B.out.resize(12);
  batch_gadget(B, batch, depth);
  uint32_t hdr[3] = {(uint32_t)B.w.size(), n_public_of(batch, depth), B.n_constraints};
  memcpy(B.out.data(), hdr, 12);
  *r1cs_bin = dup_bytes(B.out.data(), B.out.size());
  *r1cs_len = B.out.size();
  if (!*r1cs_bin) { set_error("out of memory"); return ZKR_ERR_ARG; }
  return ZKR_OK;
}

int zkr_rollup_witness(uint32_t batch, uint32_t depth, const uint8_t *inputs, size_t n_inputs, void **witness_bin, size_t *witness_len) {
  if (!inputs || !witness_bin || !witness_len) { set_error("null argument"); return ZKR_ERR_ARG; }
  int rc = check_geometry(batch, depth);
  if (rc) return rc;
  const uint32_t p = n_public_of(batch, depth);
  if (n_inputs != p - 1) { set_error("rollup circuit (%u, %u) takes %u inputs, got %zu", batch, depth, p - 1, n_inputs); return ZKR_ERR_ARG; }
  std::vector<Fr> in(n_inputs);
  for (size_t i = 0; i < n_inputs; i++)
    if (!fr_read_std(inputs + 32 * i, in[i])) { set_error("input %zu >= r", i); return ZKR_ERR_ARG; }
  std::string err;
  uint8_t *out = nullptr;
  size_t out_len = 0;
  if (!batch_witness(batch, depth, in.data(), inputs, &out, &out_len, err)) { set_error("%s", err.c_str()); return err == "out of memory" ? ZKR_ERR_ARG : ZKR_ERR_UNSATISFIED; }
  *witness_bin = out;
  *witness_len = out_len;
  return ZKR_OK;
}

}  // extern "C"
