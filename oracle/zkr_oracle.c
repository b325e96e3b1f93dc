/* zkr_oracle.c -- CPU ORACLE (plain C), TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * It is the checker for the HIP path, never the thing shipped or measured as the product.
 *
 * PARITY UNPINNED for proof bytes: the reference holds no golden proof / NTT / MSM vector
 * (SURVEY.md 8(c)); this restatement is pinned (tests/test_oracle.py) against oracle/groth16.py,
 * whose three independent h routes, naive per-signal scalar-mul prover and toxic-waste closed
 * form agree bit-for-bit and whose proofs satisfy the TxVerifier.sol pairing equation.
 *
 * Restates, for the path `wasmBn128.groth16GenProof(witnessBin, provingKeyBin)`
 * (/root/reference/operator/src/snarks/common.ts:29; websnark@0.0.5, un-vendored):
 *   input layouts   /root/reference/operator/src/utils/binarify.ts:10-48 (witness), :143-206 (key)
 *   algorithm       SURVEY.md Appendix B (published websnark/snarkjs algorithm):
 *                   QAP evaluation -> iNTT_m x2 -> coset NTT_m x2 -> pointwise -> iNTT_2m -> upper half;
 *                   five multiexps; blinding; affine; de-Montgomery.
 * 4x64-bit limbs, Montgomery form, unsigned __int128 products.  zo_prove is single-threaded; zo_prove_mt runs the
 * same algorithm on all host threads (OpenMP: the five multiexps cut into point slices, NTT butterflies and QAP
 * columns in parallel loops) for bench.py's all-cores CPU baseline -- same proof bytes.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;           /* field element (Fq or Fr by context) */
typedef struct { uint64_t p[4]; uint64_t inv; fe r1, r2; } field; /* modulus, -p^-1 mod 2^64, R, R^2 */

static const field FQ = {
  {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  0x87d20782e4866389ULL,
  {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
  {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}}};
static const field FR = {
  {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  0xc2e1f593efffffffULL,
  {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
  {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}};

static int fe_is_zero(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }
static int geq(const uint64_t *a, const uint64_t *p) {
  for (int i = 3; i >= 0; i--) { if (a[i] != p[i]) return a[i] > p[i]; }
  return 1;
}
static void sub_p(uint64_t *a, const uint64_t *p) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - p[i] - (uint64_t)br; a[i] = (uint64_t)t; br = (t >> 64) & 1; }
}
static void f_add(const field *F, fe *o, const fe *a, const fe *b) {
  u128 c = 0; fe t;
  for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; t.v[i] = (uint64_t)c; c >>= 64; }
  if (c || geq(t.v, F->p)) sub_p(t.v, F->p);
  *o = t;
}
static void f_sub(const field *F, fe *o, const fe *a, const fe *b) {
  u128 br = 0; fe t;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a->v[i] - b->v[i] - (uint64_t)br; t.v[i] = (uint64_t)d; br = (d >> 64) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t.v[i] + F->p[i]; t.v[i] = (uint64_t)c; c >>= 64; } }
  *o = t;
}
static void f_neg(const field *F, fe *o, const fe *a) { fe z = {{0, 0, 0, 0}}; f_sub(F, o, &z, a); }
/* Montgomery product a*b/2^256 mod p (CIOS) */
static void f_mul(const field *F, fe *o, const fe *a, const fe *b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t mm = t[0] * F->inv;
    c = ((u128)mm * F->p[0] + t[0]) >> 64;
    for (int j = 1; j < 4; j++) { c += (u128)mm * F->p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  fe r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || geq(r.v, F->p)) sub_p(r.v, F->p);
  *o = r;
}
static void f_sqr(const field *F, fe *o, const fe *a) { f_mul(F, o, a, a); }
static void f_to_mont(const field *F, fe *o, const fe *a) { f_mul(F, o, a, &F->r2); }
static void f_from_mont(const field *F, fe *o, const fe *a) { fe one = {{1, 0, 0, 0}}; f_mul(F, o, a, &one); }
static void f_pow(const field *F, fe *o, const fe *a, const uint64_t e[4]) {
  fe r = F->r1, b = *a;
  for (int i = 0; i < 256; i++) { if ((e[i >> 6] >> (i & 63)) & 1) f_mul(F, &r, &r, &b); f_sqr(F, &b, &b); }
  *o = r;
}
static void f_inv(const field *F, fe *o, const fe *a) {
  uint64_t e[4] = {F->p[0] - 2, F->p[1], F->p[2], F->p[3]};
  f_pow(F, o, a, e);
}

/* ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2+1) */
typedef struct { fe a, b; } fe2;
static void f2_add(fe2 *o, const fe2 *x, const fe2 *y) { f_add(&FQ, &o->a, &x->a, &y->a); f_add(&FQ, &o->b, &x->b, &y->b); }
static void f2_sub(fe2 *o, const fe2 *x, const fe2 *y) { f_sub(&FQ, &o->a, &x->a, &y->a); f_sub(&FQ, &o->b, &x->b, &y->b); }
static void f2_mul(fe2 *o, const fe2 *x, const fe2 *y) {
  fe t0, t1, s0, s1, m;
  f_mul(&FQ, &t0, &x->a, &y->a); f_mul(&FQ, &t1, &x->b, &y->b);
  f_add(&FQ, &s0, &x->a, &x->b); f_add(&FQ, &s1, &y->a, &y->b); f_mul(&FQ, &m, &s0, &s1);
  f_sub(&FQ, &o->a, &t0, &t1); f_sub(&FQ, &m, &m, &t0); f_sub(&FQ, &o->b, &m, &t1);
}
static void f2_sqr(fe2 *o, const fe2 *x) { fe2 t = *x; f2_mul(o, &t, &t); }
static int f2_is_zero(const fe2 *x) { return fe_is_zero(&x->a) && fe_is_zero(&x->b); }
static void f2_inv(fe2 *o, const fe2 *x) {
  fe n, t; f_sqr(&FQ, &n, &x->a); f_sqr(&FQ, &t, &x->b); f_add(&FQ, &n, &n, &t); f_inv(&FQ, &n, &n);
  f_mul(&FQ, &o->a, &x->a, &n); f_mul(&FQ, &t, &x->b, &n); f_neg(&FQ, &o->b, &t);
}

/* ------------------------------------------------------------------ curve groups, Jacobian, generic over the coordinate field
 * The same textbook formulas are instantiated for Fq (G1) and Fq2 (G2) through macros. */
#define DEF_GROUP(NAME, T, ADD, SUB, MUL, SQR, ISZ)                                                        \
  typedef struct { T x, y, z; } NAME##_jac;                                                                 \
  typedef struct { T x, y; int inf; } NAME##_aff;                                                           \
  static void NAME##_set_inf(NAME##_jac *p) { memset(p, 0, sizeof(*p)); }                                   \
  static int NAME##_is_inf(const NAME##_jac *p) { return ISZ(&p->z); }                                      \
  static void NAME##_dbl(NAME##_jac *o, const NAME##_jac *p) {                                              \
    if (ISZ(&p->z) || ISZ(&p->y)) { NAME##_set_inf(o); return; }                                            \
    T A, B, C, D, E, F, t, x3, y3, z3;                                                                      \
    SQR(&A, &p->x); SQR(&B, &p->y); SQR(&C, &B);                                                            \
    ADD(&t, &p->x, &B); SQR(&t, &t); SUB(&t, &t, &A); SUB(&t, &t, &C); ADD(&D, &t, &t);                     \
    ADD(&E, &A, &A); ADD(&E, &E, &A); SQR(&F, &E);                                                          \
    SUB(&x3, &F, &D); SUB(&x3, &x3, &D);                                                                    \
    SUB(&t, &D, &x3); MUL(&y3, &E, &t); ADD(&C, &C, &C); ADD(&C, &C, &C); ADD(&C, &C, &C); SUB(&y3, &y3, &C); \
    MUL(&z3, &p->y, &p->z); ADD(&z3, &z3, &z3);                                                             \
    o->x = x3; o->y = y3; o->z = z3;                                                                        \
  }                                                                                                         \
  static void NAME##_add_mixed(NAME##_jac *o, const NAME##_jac *p, const NAME##_aff *q, const T *one) {     \
    if (q->inf) { *o = *p; return; }                                                                        \
    if (ISZ(&p->z)) { o->x = q->x; o->y = q->y; o->z = *one; return; }                                      \
    T z1z1, u2, s2, h, r, hh, hhh, v, x3, y3, z3, t;                                                        \
    SQR(&z1z1, &p->z); MUL(&u2, &q->x, &z1z1); MUL(&s2, &q->y, &p->z); MUL(&s2, &s2, &z1z1);                \
    SUB(&h, &u2, &p->x); SUB(&r, &s2, &p->y);                                                               \
    if (ISZ(&h)) { if (ISZ(&r)) { NAME##_dbl(o, p); } else { NAME##_set_inf(o); } return; }                 \
    SQR(&hh, &h); MUL(&hhh, &hh, &h); MUL(&v, &p->x, &hh);                                                  \
    SQR(&x3, &r); SUB(&x3, &x3, &hhh); SUB(&x3, &x3, &v); SUB(&x3, &x3, &v);                                \
    SUB(&t, &v, &x3); MUL(&y3, &r, &t); MUL(&t, &p->y, &hhh); SUB(&y3, &y3, &t);                            \
    MUL(&z3, &p->z, &h);                                                                                    \
    o->x = x3; o->y = y3; o->z = z3;                                                                        \
  }                                                                                                         \
  static void NAME##_add(NAME##_jac *o, const NAME##_jac *p, const NAME##_jac *q) {                         \
    if (ISZ(&q->z)) { *o = *p; return; }                                                                    \
    if (ISZ(&p->z)) { *o = *q; return; }                                                                    \
    T z1z1, z2z2, u1, u2, s1, s2, h, r, hh, hhh, v, x3, y3, z3, t;                                          \
    SQR(&z1z1, &p->z); SQR(&z2z2, &q->z); MUL(&u1, &p->x, &z2z2); MUL(&u2, &q->x, &z1z1);                   \
    MUL(&s1, &p->y, &q->z); MUL(&s1, &s1, &z2z2); MUL(&s2, &q->y, &p->z); MUL(&s2, &s2, &z1z1);             \
    SUB(&h, &u2, &u1); SUB(&r, &s2, &s1);                                                                   \
    if (ISZ(&h)) { if (ISZ(&r)) { NAME##_dbl(o, p); } else { NAME##_set_inf(o); } return; }                 \
    SQR(&hh, &h); MUL(&hhh, &hh, &h); MUL(&v, &u1, &hh);                                                    \
    SQR(&x3, &r); SUB(&x3, &x3, &hhh); SUB(&x3, &x3, &v); SUB(&x3, &x3, &v);                                \
    SUB(&t, &v, &x3); MUL(&y3, &r, &t); MUL(&t, &s1, &hhh); SUB(&y3, &y3, &t);                              \
    MUL(&z3, &p->z, &q->z); MUL(&z3, &z3, &h);                                                              \
    o->x = x3; o->y = y3; o->z = z3;                                                                        \
  }

static void q_add(fe *o, const fe *a, const fe *b) { f_add(&FQ, o, a, b); }
static void q_sub(fe *o, const fe *a, const fe *b) { f_sub(&FQ, o, a, b); }
static void q_mul(fe *o, const fe *a, const fe *b) { f_mul(&FQ, o, a, b); }
static void q_sqr(fe *o, const fe *a) { f_mul(&FQ, o, a, a); }
DEF_GROUP(g1, fe, q_add, q_sub, q_mul, q_sqr, fe_is_zero)
DEF_GROUP(g2, fe2, f2_add, f2_sub, f2_mul, f2_sqr, f2_is_zero)

static void g1_to_affine(g1_aff *o, const g1_jac *p) {
  if (g1_is_inf(p)) { memset(o, 0, sizeof(*o)); o->inf = 1; return; }
  fe zi, zi2; f_inv(&FQ, &zi, &p->z); q_sqr(&zi2, &zi);
  q_mul(&o->x, &p->x, &zi2); q_mul(&zi2, &zi2, &zi); q_mul(&o->y, &p->y, &zi2); o->inf = 0;
}
static void g2_to_affine(g2_aff *o, const g2_jac *p) {
  if (g2_is_inf(p)) { memset(o, 0, sizeof(*o)); o->inf = 1; return; }
  fe2 zi, zi2; f2_inv(&zi, &p->z); f2_sqr(&zi2, &zi);
  f2_mul(&o->x, &p->x, &zi2); f2_mul(&zi2, &zi2, &zi); f2_mul(&o->y, &p->y, &zi2); o->inf = 0;
}
static const fe2 *fq2_one(void) { static fe2 one; static int init = 0; if (!init) { one.a = FQ.r1; memset(&one.b, 0, sizeof(fe)); init = 1; } return &one; }

/* wire decoding: binarify.ts writePoint writes (x, y) Montgomery; snarkjs' affine zero [0,1,0] arrives
 * as x = 0, y = mont(1) (binarify.ts:92-95); x = 0 is never on y^2 = x^3 + 3 so it is unambiguous. */
static void g1_load(g1_aff *o, const uint8_t *b) { memcpy(&o->x, b, 32); memcpy(&o->y, b + 32, 32); o->inf = fe_is_zero(&o->x); }
static void g2_load(g2_aff *o, const uint8_t *b) {
  memcpy(&o->x.a, b, 32); memcpy(&o->x.b, b + 32, 32); memcpy(&o->y.a, b + 64, 32); memcpy(&o->y.b, b + 96, 32);
  o->inf = f2_is_zero(&o->x);
}

static unsigned scalar_bits(const uint64_t *s, unsigned lo, unsigned c) {
  unsigned w = lo >> 6, sh = lo & 63;
  uint64_t v = s[w] >> sh;
  if (sh + c > 64 && w < 3) v |= s[w + 1] << (64 - sh);
  return (unsigned)(v & ((1u << c) - 1));
}

/* Pippenger bucket method, unsigned c-bit digits; scalars standard-form 256-bit LE (as in witnessBin) */
#define DEF_MSM(NAME, STRIDE, ONE)                                                                          \
  static void NAME##_msm_core(NAME##_jac *out, const uint8_t *pts, const uint8_t *scal, size_t n) {         \
    unsigned c = 1; while ((1ull << (c + 3)) < n + 16 && c < 16) c++;                                       \
    unsigned nw = (254 + c - 1) / c; size_t nb = (size_t)1 << c;                                            \
    NAME##_jac *bk = (NAME##_jac *)malloc(nb * sizeof(NAME##_jac));                                         \
    NAME##_jac total; NAME##_set_inf(&total);                                                               \
    for (int w = (int)nw - 1; w >= 0; w--) {                                                                \
      for (unsigned k = 0; k < c; k++) NAME##_dbl(&total, &total);                                          \
      memset(bk, 0, nb * sizeof(NAME##_jac));                                                               \
      for (size_t i = 0; i < n; i++) {                                                                      \
        uint64_t s[4]; memcpy(s, scal + 32 * i, 32);                                                        \
        unsigned d = scalar_bits(s, w * c, c); if (!d) continue;                                            \
        NAME##_aff q; NAME##_load(&q, pts + (size_t)STRIDE * i); if (q.inf) continue;                       \
        NAME##_add_mixed(&bk[d], &bk[d], &q, ONE);                                                          \
      }                                                                                                     \
      NAME##_jac run, sum; NAME##_set_inf(&run); NAME##_set_inf(&sum);                                      \
      for (size_t b = nb - 1; b >= 1; b--) { NAME##_add(&run, &run, &bk[b]); NAME##_add(&sum, &sum, &run); } \
      NAME##_add(&total, &total, &sum);                                                                     \
    }                                                                                                       \
    free(bk); *out = total;                                                                                 \
  }
DEF_MSM(g1, 64, &FQ.r1)
DEF_MSM(g2, 128, fq2_one())

static void g1_scalar_mul(g1_jac *o, const g1_jac *p, const uint64_t k[4]) {
  g1_jac acc; g1_set_inf(&acc);
  for (int i = 255; i >= 0; i--) { g1_dbl(&acc, &acc); if ((k[i >> 6] >> (i & 63)) & 1) g1_add(&acc, &acc, p); }
  *o = acc;
}
static void g2_scalar_mul(g2_jac *o, const g2_jac *p, const uint64_t k[4]) {
  g2_jac acc; g2_set_inf(&acc);
  for (int i = 255; i >= 0; i--) { g2_dbl(&acc, &acc); if ((k[i >> 6] >> (i & 63)) & 1) g2_add(&acc, &acc, p); }
  *o = acc;
}

/* ------------------------------------------------------------------ NTT over Fr (Montgomery inside) */
static void fr_root_of_unity(fe *o, unsigned logn) {
  /* omega_{2^28} = 5^((r-1)/2^28), then square down */
  fe five = {{5, 0, 0, 0}}, g; f_to_mont(&FR, &g, &five);
  uint64_t e[4] = {FR.p[0] - 1, FR.p[1], FR.p[2], FR.p[3]};
  /* e = (r-1) >> 28 */
  for (int i = 0; i < 4; i++) e[i] = (e[i] >> 28) | (i < 3 ? e[i + 1] << 36 : 0);
  f_pow(&FR, &g, &g, e);
  for (unsigned k = logn; k < 28; k++) f_sqr(&FR, &g, &g);
  *o = g;
}
static void ntt_mont(fe *a, unsigned logn, int inverse) {
  size_t n = (size_t)1 << logn;
  for (size_t i = 1, j = 0; i < n; i++) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  fe *tw = (fe *)malloc((n / 2 + 1) * sizeof(fe));
  for (unsigned s = 1; s <= logn; s++) {
    size_t len = (size_t)1 << s, half = len >> 1;
    fe wl; fr_root_of_unity(&wl, s); if (inverse) f_inv(&FR, &wl, &wl);
    tw[0] = FR.r1; for (size_t k = 1; k < half; k++) f_mul(&FR, &tw[k], &tw[k - 1], &wl);
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < half; k++) {
        fe u = a[i + k], v; f_mul(&FR, &v, &a[i + k + half], &tw[k]);
        f_add(&FR, &a[i + k], &u, &v); f_sub(&FR, &a[i + k + half], &u, &v);
      }
  }
  free(tw);
  if (inverse) {
    fe ni = {{n, 0, 0, 0}}; f_to_mont(&FR, &ni, &ni); f_inv(&FR, &ni, &ni);
    for (size_t i = 0; i < n; i++) f_mul(&FR, &a[i], &a[i], &ni);
  }
}

/* ------------------------------------------------------------------ exported API (ctypes) */
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

/* in/out: n = 2^logn standard-form LE elements */
void zo_ntt(uint8_t *data, unsigned logn, int inverse) {
  size_t n = (size_t)1 << logn; fe *a = (fe *)malloc(n * sizeof(fe));
  for (size_t i = 0; i < n; i++) { memcpy(&a[i], data + 32 * i, 32); f_to_mont(&FR, &a[i], &a[i]); }
  ntt_mont(a, logn, inverse);
  for (size_t i = 0; i < n; i++) { f_from_mont(&FR, &a[i], &a[i]); memcpy(data + 32 * i, &a[i], 32); }
  free(a);
}

/* points: Montgomery affine as in the key sections; scalars standard LE; out standard-form affine LE; returns 1 if infinity */
int zo_msm_g1(const uint8_t *pts, const uint8_t *scal, size_t n, uint8_t out[64]) {
  g1_jac j; g1_msm_core(&j, pts, scal, n); g1_aff a; g1_to_affine(&a, &j);
  if (a.inf) { memset(out, 0, 64); return 1; }
  f_from_mont(&FQ, &a.x, &a.x); f_from_mont(&FQ, &a.y, &a.y); memcpy(out, &a.x, 32); memcpy(out + 32, &a.y, 32); return 0;
}
int zo_msm_g2(const uint8_t *pts, const uint8_t *scal, size_t n, uint8_t out[128]) {
  g2_jac j; g2_msm_core(&j, pts, scal, n); g2_aff a; g2_to_affine(&a, &j);
  if (a.inf) { memset(out, 0, 128); return 1; }
  fe *c[4] = {&a.x.a, &a.x.b, &a.y.a, &a.y.b};
  for (int i = 0; i < 4; i++) { f_from_mont(&FQ, c[i], c[i]); memcpy(out + 32 * i, c[i], 32); }
  return 0;
}

/* a_c = sum_s polsA[s][c] w_s from the key's per-signal sparse columns (binarify.ts:104-113); Montgomery out */
static const uint8_t *qap_eval(fe *out, size_t m, const uint8_t *p, size_t n, const fe *w_mont) {
  memset(out, 0, m * sizeof(fe));
  for (size_t s = 0; s < n; s++) {
    uint32_t k = rd32(p); p += 4;
    for (uint32_t e = 0; e < k; e++) {
      uint32_t c = rd32(p); fe cf, t; memcpy(&cf, p + 4, 32); p += 36;
      f_mul(&FR, &t, &cf, &w_mont[s]); f_add(&FR, &out[c], &out[c], &t);
    }
  }
  return p;
}

/* h (standard form, m x 32 B) by the websnark route.  Returns 0 on success. */
int zo_calc_h(const uint8_t *pk, size_t pk_len, const uint8_t *witness, size_t n_w, uint8_t *h_out) {
  if (pk_len < 488) return -1;
  size_t n = rd32(pk), m = rd32(pk + 8);
  if (n_w != n || (m & (m - 1))) return -2;
  unsigned logm = 0; while (((size_t)1 << logm) < m) logm++;
  fe *w = (fe *)malloc(n * sizeof(fe));
  for (size_t i = 0; i < n; i++) { memcpy(&w[i], witness + 32 * i, 32); f_to_mont(&FR, &w[i], &w[i]); }
  fe *a = (fe *)malloc(m * sizeof(fe)), *b = (fe *)malloc(m * sizeof(fe));
  fe *ac = (fe *)malloc(m * sizeof(fe)), *bc = (fe *)malloc(m * sizeof(fe));
  fe *ev = (fe *)malloc(2 * m * sizeof(fe));
  qap_eval(a, m, pk + rd32(pk + 12), n, w);
  qap_eval(b, m, pk + rd32(pk + 16), n, w);
  memcpy(ac, a, m * sizeof(fe)); memcpy(bc, b, m * sizeof(fe));
  ntt_mont(ac, logm, 1); ntt_mont(bc, logm, 1);
  fe g, gi = FR.r1; fr_root_of_unity(&g, logm + 1);
  for (size_t i = 0; i < m; i++) { f_mul(&FR, &ac[i], &ac[i], &gi); f_mul(&FR, &bc[i], &bc[i], &gi); f_mul(&FR, &gi, &gi, &g); }
  ntt_mont(ac, logm, 0); ntt_mont(bc, logm, 0);
  for (size_t c = 0; c < m; c++) { f_mul(&FR, &ev[2 * c], &a[c], &b[c]); f_mul(&FR, &ev[2 * c + 1], &ac[c], &bc[c]); }
  ntt_mont(ev, logm + 1, 1);
  for (size_t i = 0; i < m; i++) { fe t; f_from_mont(&FR, &t, &ev[m + i]); memcpy(h_out + 32 * i, &t, 32); }
  free(w); free(a); free(b); free(ac); free(bc); free(ev);
  return 0;
}

/* full proof; r, s standard-form LE 32 B; out = pi_a(64) | pi_b(128: x.re,x.im,y.re,y.im) | pi_c(64), standard LE.
 * timings (optional, 3 doubles): seconds in [calc_h, msm, total] */
#include <time.h>
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int zo_prove(const uint8_t *pk, size_t pk_len, const uint8_t *witness, size_t n_w, const uint8_t *r32, const uint8_t *s32,
             uint8_t out[256], double *timings) {
  double t0 = now_s();
  if (pk_len < 488) return -1;
  size_t n = rd32(pk), p = rd32(pk + 4), m = rd32(pk + 8);
  if (n_w != n) return -2;
  uint8_t *h = (uint8_t *)malloc(32 * m);
  int rc = zo_calc_h(pk, pk_len, witness, n_w, h); if (rc) { free(h); return rc; }
  double t1 = now_s();
  const uint8_t *pA = pk + rd32(pk + 20), *pB1 = pk + rd32(pk + 24), *pB2 = pk + rd32(pk + 28), *pC = pk + rd32(pk + 32), *pH = pk + rd32(pk + 36);
  g1_jac A, B1, C, H; g2_jac B2;
  g1_msm_core(&A, pA, witness, n); g1_msm_core(&B1, pB1, witness, n); g2_msm_core(&B2, pB2, witness, n);
  g1_msm_core(&C, pC, witness + 32 * (p + 1), n - p - 1); g1_msm_core(&H, pH, h, m);
  double t2 = now_s();
  g1_aff alfa1, beta1, delta1; g2_aff beta2, delta2;
  g1_load(&alfa1, pk + 40); g1_load(&beta1, pk + 104); g1_load(&delta1, pk + 168); g2_load(&beta2, pk + 232); g2_load(&delta2, pk + 360);
  uint64_t r[4], s[4]; memcpy(r, r32, 32); memcpy(s, s32, 32);
  g1_jac d1j = {delta1.x, delta1.y, FQ.r1}, t, pia, pib1, pic; g2_jac d2j = {delta2.x, delta2.y, *fq2_one()}, t2j, pib;
  g1_add_mixed(&pia, &A, &alfa1, &FQ.r1); g1_scalar_mul(&t, &d1j, r); g1_add(&pia, &pia, &t);
  g2_add_mixed(&pib, &B2, &beta2, fq2_one()); g2_scalar_mul(&t2j, &d2j, s); g2_add(&pib, &pib, &t2j);
  g1_add_mixed(&pib1, &B1, &beta1, &FQ.r1); g1_scalar_mul(&t, &d1j, s); g1_add(&pib1, &pib1, &t);
  g1_add(&pic, &C, &H);
  g1_scalar_mul(&t, &pia, s); g1_add(&pic, &pic, &t);
  g1_scalar_mul(&t, &pib1, r); g1_add(&pic, &pic, &t);
  fe rm, sm, rs; memcpy(&rm, r, 32); memcpy(&sm, s, 32); f_to_mont(&FR, &rm, &rm); f_to_mont(&FR, &sm, &sm);
  f_mul(&FR, &rs, &rm, &sm); f_neg(&FR, &rs, &rs); f_from_mont(&FR, &rs, &rs);
  g1_scalar_mul(&t, &d1j, rs.v); g1_add(&pic, &pic, &t);
  g1_aff a, c; g2_aff b; g1_to_affine(&a, &pia); g2_to_affine(&b, &pib); g1_to_affine(&c, &pic);
  if (a.inf || b.inf || c.inf) { free(h); return -3; }
  fe *o[8] = {&a.x, &a.y, &b.x.a, &b.x.b, &b.y.a, &b.y.b, &c.x, &c.y};
  for (int i = 0; i < 8; i++) { f_from_mont(&FQ, o[i], o[i]); memcpy(out + 32 * i, o[i], 32); }
  free(h);
  if (timings) { timings[0] = t1 - t0; timings[1] = t2 - t1; timings[2] = now_s() - t0; }
  return 0;
}

/* ------------------------------------------------------------------ all-host-threads variant (bench.py cpu_baseline)
 * Same mathematics as zo_prove; the group sums are associative and the result is made affine, so the bytes are equal. */
#ifdef _OPENMP
#include <omp.h>
#endif
int zo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
/* in-place NTT with the butterflies of every stage split over the threads; one twiddle table w^k, k < n/2 */
static void ntt_mont_mt(fe *a, unsigned logn, int inverse) {
  size_t n = (size_t)1 << logn;
  for (size_t i = 1, j = 0; i < n; i++) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  fe w; fr_root_of_unity(&w, logn); if (inverse) f_inv(&FR, &w, &w);
  fe *tw = (fe *)malloc((n / 2 + 1) * sizeof(fe));
  /* w^k by blocks: block starts from a short serial chain, the blocks in parallel */
  const size_t half_n = n / 2 ? n / 2 : 1, BL = 4096;
  size_t nblk = (half_n + BL - 1) / BL;
  fe wb = FR.r1; { fe t = w; size_t e = BL; fe acc = FR.r1; while (e) { if (e & 1) f_mul(&FR, &acc, &acc, &t); f_sqr(&FR, &t, &t); e >>= 1; } wb = acc; }
  fe *start = (fe *)malloc(nblk * sizeof(fe));
  start[0] = FR.r1; for (size_t b = 1; b < nblk; b++) f_mul(&FR, &start[b], &start[b - 1], &wb);
#pragma omp parallel for schedule(static)
  for (long b = 0; b < (long)nblk; b++) {
    size_t lo = (size_t)b * BL, hi = lo + BL < half_n ? lo + BL : half_n;
    tw[lo] = start[b];
    for (size_t k = lo + 1; k < hi; k++) f_mul(&FR, &tw[k], &tw[k - 1], &w);
  }
  free(start);
  for (unsigned s = 1; s <= logn; s++) {
    size_t half = (size_t)1 << (s - 1), stride = (n / 2) >> (s - 1);
#pragma omp parallel for schedule(static)
    for (long q = 0; q < (long)(n / 2); q++) {
      size_t k = (size_t)q & (half - 1), i = (((size_t)q >> (s - 1)) << s) + k;
      fe u = a[i], v; f_mul(&FR, &v, &a[i + half], &tw[k * stride]);
      f_add(&FR, &a[i], &u, &v); f_sub(&FR, &a[i + half], &u, &v);
    }
  }
  free(tw);
  if (inverse) {
    fe ni = {{n, 0, 0, 0}}; f_to_mont(&FR, &ni, &ni); f_inv(&FR, &ni, &ni);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) f_mul(&FR, &a[i], &a[i], &ni);
  }
}

static int calc_h_mt(const uint8_t *pk, size_t pk_len, const uint8_t *witness, size_t n_w, uint8_t *h_out) {
  if (pk_len < 488) return -1;
  size_t n = rd32(pk), m = rd32(pk + 8);
  if (n_w != n || (m & (m - 1))) return -2;
  unsigned logm = 0; while (((size_t)1 << logm) < m) logm++;
  fe *w = (fe *)malloc(n * sizeof(fe));
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)n; i++) { memcpy(&w[i], witness + 32 * i, 32); f_to_mont(&FR, &w[i], &w[i]); }
  fe *a = (fe *)malloc(m * sizeof(fe)), *b = (fe *)malloc(m * sizeof(fe));
  fe *ac = (fe *)malloc(m * sizeof(fe)), *bc = (fe *)malloc(m * sizeof(fe));
  fe *ev = (fe *)malloc(2 * m * sizeof(fe));
#pragma omp parallel sections
  {
#pragma omp section
    qap_eval(a, m, pk + rd32(pk + 12), n, w);   /* column-major scatter: one thread per matrix */
#pragma omp section
    qap_eval(b, m, pk + rd32(pk + 16), n, w);
  }
  memcpy(ac, a, m * sizeof(fe)); memcpy(bc, b, m * sizeof(fe));
  ntt_mont_mt(ac, logm, 1); ntt_mont_mt(bc, logm, 1);
  fe g; fr_root_of_unity(&g, logm + 1);
  {  /* coset shift g^i: blocks in parallel */
    const size_t BL = 4096; size_t nblk = (m + BL - 1) / BL;
    fe gb = FR.r1, t = g; size_t e = BL; while (e) { if (e & 1) f_mul(&FR, &gb, &gb, &t); f_sqr(&FR, &t, &t); e >>= 1; }
    fe *start = (fe *)malloc(nblk * sizeof(fe));
    start[0] = FR.r1; for (size_t k = 1; k < nblk; k++) f_mul(&FR, &start[k], &start[k - 1], &gb);
#pragma omp parallel for schedule(static)
    for (long q = 0; q < (long)nblk; q++) {
      size_t lo = (size_t)q * BL, hi = lo + BL < m ? lo + BL : m;
      fe gi = start[q];
      for (size_t i = lo; i < hi; i++) { f_mul(&FR, &ac[i], &ac[i], &gi); f_mul(&FR, &bc[i], &bc[i], &gi); f_mul(&FR, &gi, &gi, &g); }
    }
    free(start);
  }
  ntt_mont_mt(ac, logm, 0); ntt_mont_mt(bc, logm, 0);
#pragma omp parallel for schedule(static)
  for (long c = 0; c < (long)m; c++) { f_mul(&FR, &ev[2 * c], &a[c], &b[c]); f_mul(&FR, &ev[2 * c + 1], &ac[c], &bc[c]); }
  ntt_mont_mt(ev, logm + 1, 1);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)m; i++) { fe t; f_from_mont(&FR, &t, &ev[m + i]); memcpy(h_out + 32 * i, &t, 32); }
  free(w); free(a); free(b); free(ac); free(bc); free(ev);
  return 0;
}

/* one multiexp as `slices` independent Pippenger runs over point ranges (each picks its own window size) */
typedef struct { int g2; const uint8_t *pts, *scal; size_t n; void *out; } msm_task;
int zo_prove_mt(const uint8_t *pk, size_t pk_len, const uint8_t *witness, size_t n_w, const uint8_t *r32, const uint8_t *s32,
                uint8_t out[256], double *timings, int threads) {
  double t0 = now_s();
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
  int T = omp_get_max_threads();
#else
  int T = 1; (void)threads;
#endif
  if (pk_len < 488) return -1;
  size_t n = rd32(pk), p = rd32(pk + 4), m = rd32(pk + 8);
  if (n_w != n) return -2;
  uint8_t *h = (uint8_t *)malloc(32 * m);
  int rc = calc_h_mt(pk, pk_len, witness, n_w, h); if (rc) { free(h); return rc; }
  double t1 = now_s();
  const uint8_t *pA = pk + rd32(pk + 20), *pB1 = pk + rd32(pk + 24), *pB2 = pk + rd32(pk + 28), *pC = pk + rd32(pk + 32), *pH = pk + rd32(pk + 36);
  /* tasks: [A, B1, C, H] x S slices of G1 work, B2 x 3S slices (a G2 addition costs about three G1 additions) */
  int S = T < 1 ? 1 : T;
  while ((size_t)S * 1024 > n && S > 1) S /= 2;  /* keep slices above ~1000 points */
  struct { int g2; const uint8_t *pts, *scal; size_t cnt; int slices; } ms[5] = {
    {0, pA, witness, n, S}, {0, pB1, witness, n, S}, {1, pB2, witness, n, 3 * S}, {0, pC, witness + 32 * (p + 1), n - p - 1, S}, {0, pH, h, m, S}};
  int ntask = 0, first[6];
  for (int i = 0; i < 5; i++) { first[i] = ntask; ntask += ms[i].slices; }
  first[5] = ntask;
  g1_jac *r1 = (g1_jac *)calloc(ntask, sizeof(g1_jac));
  g2_jac *r2 = (g2_jac *)calloc(ntask, sizeof(g2_jac));
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < ntask; t++) {
    int i = 0; while (t >= first[i + 1]) i++;
    int q = t - first[i];
    size_t lo = ms[i].cnt * (size_t)q / ms[i].slices, hi = ms[i].cnt * (size_t)(q + 1) / ms[i].slices;
    if (ms[i].g2) g2_msm_core(&r2[t], ms[i].pts + 128 * lo, ms[i].scal + 32 * lo, hi - lo);
    else g1_msm_core(&r1[t], ms[i].pts + 64 * lo, ms[i].scal + 32 * lo, hi - lo);
  }
  g1_jac A, B1, C, H; g2_jac B2;
  g1_jac *dst[5] = {&A, &B1, NULL, &C, &H};
  for (int i = 0; i < 5; i++) {
    if (ms[i].g2) { g2_set_inf(&B2); for (int t = first[i]; t < first[i + 1]; t++) g2_add(&B2, &B2, &r2[t]); }
    else { g1_set_inf(dst[i]); for (int t = first[i]; t < first[i + 1]; t++) g1_add(dst[i], dst[i], &r1[t]); }
  }
  free(r1); free(r2);
  double t2 = now_s();
  g1_aff alfa1, beta1, delta1; g2_aff beta2, delta2;
  g1_load(&alfa1, pk + 40); g1_load(&beta1, pk + 104); g1_load(&delta1, pk + 168); g2_load(&beta2, pk + 232); g2_load(&delta2, pk + 360);
  uint64_t r[4], s[4]; memcpy(r, r32, 32); memcpy(s, s32, 32);
  g1_jac d1j = {delta1.x, delta1.y, FQ.r1}, t, pia, pib1, pic; g2_jac d2j = {delta2.x, delta2.y, *fq2_one()}, t2j, pib;
  g1_add_mixed(&pia, &A, &alfa1, &FQ.r1); g1_scalar_mul(&t, &d1j, r); g1_add(&pia, &pia, &t);
  g2_add_mixed(&pib, &B2, &beta2, fq2_one()); g2_scalar_mul(&t2j, &d2j, s); g2_add(&pib, &pib, &t2j);
  g1_add_mixed(&pib1, &B1, &beta1, &FQ.r1); g1_scalar_mul(&t, &d1j, s); g1_add(&pib1, &pib1, &t);
  g1_add(&pic, &C, &H);
  g1_scalar_mul(&t, &pia, s); g1_add(&pic, &pic, &t);
  g1_scalar_mul(&t, &pib1, r); g1_add(&pic, &pic, &t);
  fe rm, sm, rs; memcpy(&rm, r, 32); memcpy(&sm, s, 32); f_to_mont(&FR, &rm, &rm); f_to_mont(&FR, &sm, &sm);
  f_mul(&FR, &rs, &rm, &sm); f_neg(&FR, &rs, &rs); f_from_mont(&FR, &rs, &rs);
  g1_scalar_mul(&t, &d1j, rs.v); g1_add(&pic, &pic, &t);
  g1_aff a, c; g2_aff b; g1_to_affine(&a, &pia); g2_to_affine(&b, &pib); g1_to_affine(&c, &pic);
  if (a.inf || b.inf || c.inf) { free(h); return -3; }
  fe *o[8] = {&a.x, &a.y, &b.x.a, &b.x.b, &b.y.a, &b.y.b, &c.x, &c.y};
  for (int i = 0; i < 8; i++) { f_from_mont(&FQ, o[i], o[i]); memcpy(out + 32 * i, o[i], 32); }
  free(h);
  if (timings) { timings[0] = t1 - t0; timings[1] = t2 - t1; timings[2] = now_s() - t0; timings[3] = (double)T; }
  return 0;
}

/* sum_i a_i b_i mod r over standard-form 32-byte LE elements (the dot products of the toxic-waste closed form at full
 * size, where a Python loop over 2^24 big integers takes minutes) */
void zo_fr_dot(const uint8_t *a_std, const uint8_t *b_std, size_t n, uint8_t out[32]) {
  fe acc = {{0, 0, 0, 0}};
#pragma omp parallel
  {
    fe part = {{0, 0, 0, 0}};
#pragma omp for schedule(static) nowait
    for (long i = 0; i < (long)n; i++) {
      fe x, y, t; memcpy(&x, a_std + 32 * i, 32); memcpy(&y, b_std + 32 * i, 32);
      f_to_mont(&FR, &x, &x); f_mul(&FR, &t, &x, &y);   /* (a R)(b)/R = a b */
      f_add(&FR, &part, &part, &t);
    }
#pragma omp critical
    f_add(&FR, &acc, &acc, &part);
  }
  memcpy(out, &acc, 32);
}

/* field self-test hooks */
void zo_fq_mul_std(const uint8_t a[32], const uint8_t b[32], uint8_t o[32]) {
  fe x, y; memcpy(&x, a, 32); memcpy(&y, b, 32); f_to_mont(&FQ, &x, &x); f_to_mont(&FQ, &y, &y); f_mul(&FQ, &x, &x, &y); f_from_mont(&FQ, &x, &x); memcpy(o, &x, 32);
}
void zo_fr_mul_std(const uint8_t a[32], const uint8_t b[32], uint8_t o[32]) {
  fe x, y; memcpy(&x, a, 32); memcpy(&y, b, 32); f_to_mont(&FR, &x, &x); f_to_mont(&FR, &y, &y); f_mul(&FR, &x, &x, &y); f_from_mont(&FR, &x, &x); memcpy(o, &x, 32);
}
