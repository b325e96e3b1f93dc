// rollup_witness.hpp -- the VALUE side of the rollup circuit's gadget program (rollup.cpp), written once for device and host:
// what one thread of the GPU witness builder runs per transaction (rollup_gpu.hip, zkr_rollup_witness_batch_device), and what
// the CPU test hook runs to compare it with the host Builder signal for signal without a GPU (tests/test_rollup.py).
// Reference: Circuit.calculateWitness at /root/reference/operator/src/snarks/common.ts:15-17 for
// prover/circuits/processtx.circom:10-193 (over eddsa.circom:12-139, merkletree.circom:5-84, hasher.circom:3-30).
#pragma once
#include "field.hpp"

#if defined(__HIPCC__) || defined(__HIP__)
#define ZKR_HDF __host__ __device__ static
#else
#define ZKR_HDF static inline
#endif

namespace zkr {
constexpr int MIMC_ROUNDS = 220;  // hasher.circom:8

enum TxStmt : uint32_t {
  ST_OK = 0, ST_INPUT_RANGE, ST_SIDX, ST_RIDX, ST_SBITS, ST_SORDER, ST_HCANON, ST_EDX, ST_EDY, ST_SMALL_ORDER, ST_SIGNATURE, ST_NONCE,
  ST_AMOUNT_BITS, ST_FEE_BITS, ST_BALANCE_BITS, ST_AMOUNT_POS, ST_FEE_POS, ST_CMP_BITS, ST_BALANCE_GT, ST_SLEAF, ST_RLEAF, ST_IROOT, ST_CHAIN, ST_COUNT
};
static const char *const TX_STMT_TEXT[ST_COUNT] = {  // rollup.cpp's wording
    "", "an input is not below the field modulus", "sender index fits the tree", "recipient index fits the tree", "S fits 253 bits",
    "S below the subgroup order", "hash bits are the canonical ones", "edwards x", "edwards y", "A is not a small-order point",
    "transaction signature is valid", "nonce is the sender's nonce + 1", "amount fits 250 bits", "fee fits 250 bits",
    "sender balance fits 250 bits", "amount > 0", "fee > 0", "balance comparison", "sender balance > amount + fee",
    "sender leaf is in the balance tree", "recipient leaf is in the balance tree", "intermediate root after the sender update",
    "a transaction starts from the root the previous one produced"};

constexpr int WS_PTS = 512;  // projective points of one scalar-multiplication chain (2 x 253 at most)
constexpr int TX_WS_ELEMS = 8 * WS_PTS;  // scratch of one transaction: four coordinate arrays for each of the two parts with such chains
struct TxConsts {
  Fr a, d;
  uint32_t suborder_m1[8];
  const Fr *b8x, *b8y;  // 2^i BASE8, i < 253 (affine, Montgomery)
  const Fr *mimc;       // the 220 round constants (Montgomery)
};
struct TxB {
  Fr *w;         // this transaction's private signals, in allocation order (Montgomery until the layout kernel)
  uint32_t n;    // signals allocated so far
  uint32_t cap;  // signals `w` has room for (the structure pass's count K); a program that ran past it fails with ST_COUNT
  uint32_t err;  // first violated statement in program order (TxStmt)
  Fr *wx, *wy, *wz, *wp;  // workspace: WS_PTS coordinates each
  TxConsts k;
  ZKR_HD Fr put(const Fr &v) {
#ifndef __HIP_DEVICE_COMPILE__
    // host build (zkr_rollup_witness, the CPU tests): never past the slice -- one signal too many must not write into the next
    // transaction's; the device build keeps the bare store (the program is the same: what the host run proves holds there)
    if (n >= cap) { fail(ST_COUNT); n++; return v; }
#endif
    w[n++] = v;
    return v;
  }
  ZKR_HD void fail(uint32_t code);
};
ZKR_HD void TxB::fail(uint32_t code) { if (!err) err = code; }
struct PtD { Fr x, y, z; };
ZKR_HDF PtD ptd_add(const TxB &b, const PtD &p, const PtD &q) {  // add-2008-bbjlp, complete on BabyJub (rollup.cpp pt_add)
  Fr A = mul(p.z, q.z), B = sqr(A), C = mul(p.x, q.x), D = mul(p.y, q.y);
  Fr E = mul(b.k.d, mul(C, D)), F = sub(B, E), G = add(B, E);
  Fr x3 = mul(mul(A, F), sub(sub(mul(add(p.x, p.y), add(q.x, q.y)), C), D));
  Fr y3 = mul(mul(A, G), sub(D, mul(b.k.a, C)));
  return PtD{x3, y3, mul(F, G)};
}
// affine coordinates of the cnt points in the workspace, one field inversion (rollup.cpp affine_all / batch_inverse)
ZKR_HDF void ws_affine(TxB &b, uint32_t cnt) {
  Fr acc = Fr::one();
  for (uint32_t i = 0; i < cnt; i++) {
    b.wp[i] = acc;
    Fr z = b.wz[i];
    if (!z.is_zero()) acc = mul(acc, z);
  }
  Fr ia = inv(acc);
  for (uint32_t i = cnt; i-- > 0;) {
    Fr z = b.wz[i];
    Fr zi = Fr::zero();
    if (!z.is_zero()) {
      zi = mul(ia, b.wp[i]);
      ia = mul(ia, z);
    }
    b.wx[i] = mul(b.wx[i], zi);
    b.wy[i] = mul(b.wy[i], zi);
  }
}
ZKR_HDF void tx_feistel(TxB &b, Fr &xl, Fr &xr) {  // feistel_gadget: t2, t4, x' per round
  for (int i = 0; i < MIMC_ROUNDS; i++) {
    Fr t = add(xl, b.k.mimc[i]);
    Fr t2 = b.put(sqr(t));
    Fr t4 = b.put(sqr(t2));
    Fr nx = b.put(add(xr, mul(t4, t)));
    if (i < MIMC_ROUNDS - 1) {
      xr = xl;
      xl = nx;
    } else {
      xr = nx;
    }
  }
}
ZKR_HDF Fr tx_multihash(TxB &b, const Fr *in, int n) {
  Fr r = Fr::zero(), c = Fr::zero();
  for (int i = 0; i < n; i++) {
    r = add(r, in[i]);
    tx_feistel(b, r, c);
  }
  return r;
}
ZKR_HDF uint32_t word_bit(const uint32_t (&w)[8], int i) { return i < 256 ? (w[i >> 5] >> (i & 31)) & 1u : 0u; }
// Num2Bits(n): n bit signals; the closing constraint holds exactly when the value has no bit at or above n
ZKR_HDF void tx_num2bits(TxB &b, const Fr &v, int n, uint32_t code, uint32_t (&bits)[8]) {
  Fr s = from_mont(v);
#pragma unroll
  for (int i = 0; i < 8; i++) bits[i] = s.v[i];
  const Fr one = Fr::one(), zero = Fr::zero();
  for (int i = 0; i < n; i++) b.put(word_bit(bits, i) ? one : zero);
  bool fits = true;
  for (int i = n; i < 256; i++) fits = fits && !word_bit(bits, i);
  if (!fits) b.fail(code);
  for (int i = n; i < 256; i++) bits[i >> 5] &= ~(1u << (i & 31));  // the bit signals are the low n bits
}
// bits > constant ? 1 : 0 (bits_gt_const): the running value is boolean, one product signal per bit once it is not the constant 0
ZKR_HDF uint32_t tx_bits_gt_const(TxB &b, const uint32_t (&bits)[8], int n, const uint32_t (&c)[8]) {
  uint32_t gt = 0;
  bool is_const_zero = true;
  const Fr one = Fr::one(), zero = Fr::zero();
  for (int i = 0; i < n; i++) {
    const uint32_t cb = word_bit(c, i), bi = word_bit(bits, i);
    if (cb == 0) {
      if (is_const_zero) {
        gt = bi;
        is_const_zero = false;
      } else {
        const uint32_t m = bi & gt;
        b.put(m ? one : zero);
        gt = bi | gt;
      }
    } else if (!is_const_zero) {
      gt = bi & gt;
      b.put(gt ? one : zero);
    }
  }
  return gt;
}
struct PtA { Fr x, y; };
// edwards_add: beta, gamma, delta, tau, x, y; `known` = the sum's affine coordinates when the chain has them already
ZKR_HDF __attribute__((noinline)) PtA tx_edwards_add(TxB &b, const PtA p, const PtA q, const Fr *kx = nullptr, const Fr *ky = nullptr) {
  Fr beta = b.put(mul(p.x, q.y));
  Fr gamma = b.put(mul(p.y, q.x));
  Fr delta = b.put(mul(sub(p.y, mul(p.x, b.k.a)), add(q.x, q.y)));
  Fr tau = b.put(mul(beta, gamma));
  Fr dt = mul(tau, b.k.d);
  Fr dx = add(Fr::one(), dt), dy = sub(Fr::one(), dt);
  Fr nx = add(beta, gamma), ny = sub(add(delta, mul(beta, b.k.a)), gamma);
  Fr xv, yv;
  if (kx) {
    xv = *kx, yv = *ky;
  } else {
    Fr di = inv(mul(dx, dy));
    xv = mul(nx, mul(di, dy)), yv = mul(ny, mul(di, dx));
  }
  b.put(xv);
  b.put(yv);
  if (!(mul(dx, xv) == nx)) b.fail(ST_EDX);
  if (!(mul(dy, yv) == ny)) b.fail(ST_EDY);
  return PtA{xv, yv};
}
// sum_i e_i 2^i P for a variable point (scalar_mul_any): the chain in projective coordinates first, one inversion, then the signals
ZKR_HDF PtA tx_scalar_mul_any(TxB &b, const uint32_t (&e)[8], int n, PtA q) {
  {
    PtD qq{q.x, q.y, Fr::one()}, acc = word_bit(e, 0) ? qq : PtD{Fr::zero(), Fr::one(), Fr::one()};
    for (int i = 1; i < n; i++) {
      qq = ptd_add(b, qq, qq);
      if (word_bit(e, i)) acc = ptd_add(b, acc, qq);
      b.wx[2 * (i - 1)] = acc.x, b.wy[2 * (i - 1)] = acc.y, b.wz[2 * (i - 1)] = acc.z;
      b.wx[2 * (i - 1) + 1] = qq.x, b.wy[2 * (i - 1) + 1] = qq.y, b.wz[2 * (i - 1) + 1] = qq.z;
    }
  }
  ws_affine(b, 2 * (uint32_t)(n - 1));
  PtA acc{Fr::zero(), Fr::one()};
  const Fr one = Fr::one();
  for (int i = 0; i < n; i++) {
    const bool ei = word_bit(e, i);
    // e_i * q.x and e_i * (q.y - 1) + 1
    PtA sel{b.put(ei ? q.x : Fr::zero()), Fr::zero()};
    sel.y = add(b.put(ei ? sub(q.y, one) : Fr::zero()), one);
    if (i == 0) acc = sel;
    else acc = tx_edwards_add(b, acc, sel, &b.wx[2 * (i - 1)], &b.wy[2 * (i - 1)]);
    if (i + 1 < n) q = tx_edwards_add(b, q, q, &b.wx[2 * i + 1], &b.wy[2 * i + 1]);
  }
  return acc;
}
// sum_i e_i 2^i BASE8 (scalar_mul_base8): constant multiples, linear selection, one addition per bit
ZKR_HDF PtA tx_scalar_mul_base8(TxB &b, const uint32_t (&e)[8], int n) {
  {
    PtD acc = word_bit(e, 0) ? PtD{b.k.b8x[0], b.k.b8y[0], Fr::one()} : PtD{Fr::zero(), Fr::one(), Fr::one()};
    for (int i = 1; i < n; i++) {
      if (word_bit(e, i)) acc = ptd_add(b, acc, PtD{b.k.b8x[i], b.k.b8y[i], Fr::one()});
      b.wx[i - 1] = acc.x, b.wy[i - 1] = acc.y, b.wz[i - 1] = acc.z;
    }
  }
  ws_affine(b, (uint32_t)(n - 1));
  PtA acc{Fr::zero(), Fr::one()};
  const Fr one = Fr::one();
  for (int i = 0; i < n; i++) {
    const bool ei = word_bit(e, i);
    PtA sel{ei ? b.k.b8x[i] : Fr::zero(), ei ? b.k.b8y[i] : one};  // e_i x_i, e_i (y_i - 1) + 1
    if (i == 0) acc = sel;
    else acc = tx_edwards_add(b, acc, sel, &b.wx[i - 1], &b.wy[i - 1]);
  }
  return acc;
}
ZKR_HDF bool tx_is_zero(TxB &b, const Fr &v) {  // IsZero: inverse, flag
  const bool z = v.is_zero();
  b.put(z ? Fr::zero() : inv(v));
  b.put(z ? Fr::one() : Fr::zero());
  return z;
}
// EdDSAMiMCSpongeVerifierPatched (eddsa_verify_gadget): valid ? 1 : 0
// signals of the fixed-base multiplication S * BASE8 (252 additions), of the three IsZero behind it, and of what ProcessTx
// allocates after the signature (three 250-bit range checks, two IsZero, the 252-bit comparison): where the multiplication starts
// is counted back from the first leaf hash
constexpr uint32_t TX_BASE8_SIGNALS = 252 * 6, TX_SIGZ_SIGNALS = 6, TX_TAIL_SIGNALS = 3 * 250 + 4 + 252;
// skip_base8: everything up to R8 + h * 8A; the multiplication S * BASE8 and the comparison of the two sides are left to another
// part (tx_process part 1) and to tx_finish, their signals are stepped over
ZKR_HDF bool tx_eddsa_verify(TxB &b, const Fr &ax, const Fr &ay, const Fr &S, const Fr &r8x, const Fr &r8y, const Fr &M, uint32_t o_base8, bool skip_base8) {
  uint32_t sbits[8], hbits[8];
  tx_num2bits(b, S, 253, ST_SBITS, sbits);
  if (tx_bits_gt_const(b, sbits, 253, b.k.suborder_m1)) b.fail(ST_SORDER);
  const Fr hin[5] = {r8x, r8y, ax, ay, M};
  Fr h = tx_multihash(b, hin, 5);
  tx_num2bits(b, h, 254, ST_OK, hbits);  // 254 bits always hold a field element
  uint32_t rm1[8];
#pragma unroll
  for (int i = 0; i < 8; i++) rm1[i] = FrParams::P[i];
  rm1[0] -= 1;
  if (tx_bits_gt_const(b, hbits, 254, rm1)) b.fail(ST_HCANON);
  PtA a1{ax, ay};
  PtA a2 = tx_edwards_add(b, a1, a1);
  PtA a4 = tx_edwards_add(b, a2, a2);
  PtA a8 = tx_edwards_add(b, a4, a4);
  if (tx_is_zero(b, a4.x)) b.fail(ST_SMALL_ORDER);
  PtA right2 = tx_scalar_mul_any(b, hbits, 254, a8);
  PtA right = tx_edwards_add(b, PtA{r8x, r8y}, right2);
  if (!b.err && b.n != o_base8) b.fail(ST_COUNT);  // internal: the count back from the leaf hashes is not the program's
  if (skip_base8) {
    b.n += TX_BASE8_SIGNALS + TX_SIGZ_SIGNALS;
    return true;
  }
  PtA left = tx_scalar_mul_base8(b, sbits, 253);
  const bool ex = tx_is_zero(b, sub(left.x, right.x)), ey = tx_is_zero(b, sub(left.y, right.y));
  Fr two = add(Fr::one(), Fr::one());
  Fr exy = sub(add(ex ? Fr::one() : Fr::zero(), ey ? Fr::one() : Fr::zero()), two);
  return tx_is_zero(b, exy);
}
// MerkleTreeRootConstructor (merkle_root_gadget)
ZKR_HDF Fr tx_merkle_root(TxB &b, Fr cur, const Fr *path, const uint32_t (&idx)[8], uint32_t depth) {
  for (uint32_t i = 0; i < depth; i++) {
    Fr m = b.put(word_bit(idx, (int)i) ? sub(path[i], cur) : Fr::zero());  // idx_i * (sibling - cur)
    Fr left = add(cur, m);
    Fr right = sub(add(path[i], cur), left);
    const Fr lr[2] = {left, right};
    cur = tx_multihash(b, lr, 2);
  }
  return cur;
}
constexpr uint32_t TX_MAX_DEPTH = 32;
struct TxInD {
  Fr root, tx[8], spk[2], sbal, snonce, rpk[2], rbal, rnonce, iroot;
  Fr spath[TX_MAX_DEPTH], rpath[TX_MAX_DEPTH], ipath[TX_MAX_DEPTH];
};
// ProcessTx (process_tx_gadget): returns the new root
// The program in six parts that share no signal, each a chain of dependent multiplications:
//   0  indices, message hash, signature up to R8 + h * 8A (two sponges and the variable-base multiplication: ~16 000), range checks
//   1  the fixed-base multiplication S * BASE8 (~4 300)
//   2  the sender's leaf and path        3  the sender's new leaf and path           (~10 500 each)
//   4  the recipient's leaf and path     5  the recipient's new leaf and path = the transaction's root
// and tx_finish, which compares the two sides of the signature equation once parts 0 and 1 are done (three IsZero, no inversion when
// the signature holds).  part < 0 runs everything in order (tests, the gadget comparison); the GPU builder gives each part of 64
// transactions its own wavefront (rollup_gpu.hip), the host builder's fast path a task.  Where a part starts is a matter of
// counting: a leaf hash is 4 x 660 signals, a path depth x (1 + 2 x 660), the recipient selectors 4.
constexpr uint32_t TX_PARTS = 6;
ZKR_HDF Fr tx_process(TxB &b, const TxInD &in, uint32_t depth, int part, uint32_t K) {
  const bool all = part < 0;
  const uint32_t leaf = 4 * 3 * MIMC_ROUNDS, path = depth * (1 + 2 * 3 * MIMC_ROUNDS);
  const uint32_t o_sleaf = K - (4 * leaf + 4 * path + 4), o_rleaf = o_sleaf + leaf, o_spath = o_rleaf + leaf, o_rpath = o_spath + path;
  const uint32_t o_nsleaf = o_rpath + path, o_same = o_nsleaf + leaf, o_nrleaf = o_same + 4, o_nspath = o_nrleaf + leaf, o_nrpath = o_nspath + path;
  const uint32_t o_base8 = o_sleaf - TX_TAIL_SIGNALS - TX_SIGZ_SIGNALS - TX_BASE8_SIGNALS;
  uint32_t sidx[8], ridx[8], tmp[8];
  if (all || part == 0) {
    tx_num2bits(b, in.tx[0], (int)depth, ST_SIDX, sidx);
    tx_num2bits(b, in.tx[1], (int)depth, ST_RIDX, ridx);
    const Fr m5[5] = {in.tx[0], in.tx[1], in.tx[2], in.tx[3], in.tx[4]};
    Fr msg = tx_multihash(b, m5, 5);
    if (!tx_eddsa_verify(b, in.spk[0], in.spk[1], in.tx[7], in.tx[5], in.tx[6], msg, o_base8, !all)) b.fail(ST_SIGNATURE);
    if (!(in.tx[4] == add(in.snonce, Fr::one()))) b.fail(ST_NONCE);
    tx_num2bits(b, in.tx[2], 250, ST_AMOUNT_BITS, tmp);
    tx_num2bits(b, in.tx[3], 250, ST_FEE_BITS, tmp);
    tx_num2bits(b, in.sbal, 250, ST_BALANCE_BITS, tmp);
    if (tx_is_zero(b, in.tx[2])) b.fail(ST_AMOUNT_POS);
    if (tx_is_zero(b, in.tx[3])) b.fail(ST_FEE_POS);
    {  // balance > amount + fee: bit 251 of balance - (amount + fee) - 1 + 2^251 (greater_than_250)
      Fr p251 = Fr::one();
      for (int i = 0; i < 251; i++) p251 = dbl(p251);
      Fr v = add(sub(sub(in.sbal, add(in.tx[2], in.tx[3])), Fr::one()), p251);
      tx_num2bits(b, v, 252, ST_CMP_BITS, tmp);
      if (!word_bit(tmp, 251)) b.fail(ST_BALANCE_GT);
    }
    if (!b.err && b.n != o_sleaf) b.fail(ST_COUNT);  // internal: the count above is not the structure pass's
  } else {  // the index bits without their signals
    Fr s0 = from_mont(in.tx[0]), s1 = from_mont(in.tx[1]);
#pragma unroll
    for (int i = 0; i < 8; i++) sidx[i] = s0.v[i], ridx[i] = s1.v[i];
  }
  if (part == 1) {  // S's bits without their signals (part 0 allocates them and refuses an S of more than 253 bits)
    uint32_t sbits[8];
    Fr s = from_mont(in.tx[7]);
#pragma unroll
    for (int i = 0; i < 8; i++) sbits[i] = s.v[i];
    sbits[7] &= (1u << (253 - 224)) - 1u;
    b.n = o_base8;
    tx_scalar_mul_base8(b, sbits, 253);
    if (!b.err && b.n != o_base8 + TX_BASE8_SIGNALS) b.fail(ST_COUNT);
  }
  const Fr nsbal = sub(sub(in.sbal, in.tx[2]), in.tx[3]);
  Fr root = Fr::zero();
  if (all || part == 2) {
    const Fr sl[4] = {in.spk[0], in.spk[1], in.sbal, in.snonce};
    b.n = o_sleaf;
    Fr sleaf = tx_multihash(b, sl, 4);
    b.n = o_spath;
    if (!(tx_merkle_root(b, sleaf, in.spath, sidx, depth) == in.root)) b.fail(ST_SLEAF);
  }
  if (all || part == 3) {
    const Fr nsl[4] = {in.spk[0], in.spk[1], nsbal, in.tx[4]};
    b.n = o_nsleaf;
    Fr nsleaf = tx_multihash(b, nsl, 4);
    b.n = o_nspath;
    if (!(tx_merkle_root(b, nsleaf, in.spath, sidx, depth) == in.iroot)) b.fail(ST_IROOT);
  }
  if (all || part == 4) {
    const Fr rl[4] = {in.rpk[0], in.rpk[1], in.rbal, in.rnonce};
    b.n = o_rleaf;
    Fr rleaf = tx_multihash(b, rl, 4);
    b.n = o_rpath;
    if (!(tx_merkle_root(b, rleaf, in.rpath, ridx, depth) == in.root)) b.fail(ST_RLEAF);
  }
  if (all || part == 5) {
    b.n = o_same;
    const bool same = tx_is_zero(b, sub(in.tx[0], in.tx[1]));
    Fr selbal = add(in.rbal, b.put(same ? sub(nsbal, in.rbal) : Fr::zero()));
    Fr selnonce = add(in.rnonce, b.put(same ? sub(in.tx[4], in.rnonce) : Fr::zero()));
    const Fr nrl[4] = {in.rpk[0], in.rpk[1], add(selbal, in.tx[2]), selnonce};
    Fr nrleaf = tx_multihash(b, nrl, 4);
    b.n = o_nrpath;
    root = tx_merkle_root(b, nrleaf, in.ipath, ridx, depth);
    if (!b.err && b.n != K) b.fail(ST_COUNT);
  }
  return root;
}
// What is left of the signature check when the parts ran apart: the two sides of S * BASE8 == R8 + h * 8A are the last signals of
// part 1's and of part 0's additions; their comparison is three IsZero (an inversion only where a difference is not zero, i.e.
// never for a valid signature).  w: the transaction's private signals (Montgomery).  Returns ST_SIGNATURE or 0.
ZKR_HDF uint32_t tx_finish(Fr *w, uint32_t depth, uint32_t K) {
  const uint32_t leaf = 4 * 3 * MIMC_ROUNDS, path = depth * (1 + 2 * 3 * MIMC_ROUNDS);
  const uint32_t o_sleaf = K - (4 * leaf + 4 * path + 4);
  const uint32_t o_base8 = o_sleaf - TX_TAIL_SIGNALS - TX_SIGZ_SIGNALS - TX_BASE8_SIGNALS;
  TxB b;
  b.w = w;
  b.n = o_base8 + TX_BASE8_SIGNALS;
  b.cap = K;
  b.err = ST_OK;
  const Fr rx = w[o_base8 - 2], ry = w[o_base8 - 1], lx = w[b.n - 2], ly = w[b.n - 1];
  const bool ex = tx_is_zero(b, sub(lx, rx)), ey = tx_is_zero(b, sub(ly, ry));
  Fr two = add(Fr::one(), Fr::one());
  Fr exy = sub(add(ex ? Fr::one() : Fr::zero(), ey ? Fr::one() : Fr::zero()), two);
  return tx_is_zero(b, exy) ? (uint32_t)ST_OK : (uint32_t)ST_SIGNATURE;
}
// standard-form input below r -> Montgomery; false when it is not a field element
ZKR_HDF bool tx_read_input(const Fr *p, Fr &out) {
  Fr a = *p;
  bool ge = true;
  for (int i = 7; i >= 0; i--)
    if (a.v[i] != FrParams::P[i]) { ge = a.v[i] > FrParams::P[i]; break; }
  out = to_mont(a);
  return !ge;
}

// the signal layout of BatchProcessTx(batch, depth): circom's order (rollup.cpp Layout) and the inputs of transaction i
struct TxLayout {
  uint32_t p, o_root, o_tx, o_spk, o_sbal, o_snonce, o_spath, o_rpk, o_rbal, o_rnonce, o_rpath, o_iroot, o_ipath;
};
ZKR_HDF TxLayout tx_layout(uint32_t batch, uint32_t depth) {
  TxLayout L;
  L.p = 1 + batch * (8 + 2 + 1 + 1 + 2 + 1 + 1 + 1 + 1 + 3 * depth);
  uint32_t at = 2;
  auto take = [&](uint32_t count) { uint32_t first = at; at += batch * count; return first; };
  L.o_root = take(1), L.o_tx = take(8), L.o_spk = take(2), L.o_sbal = take(1), L.o_snonce = take(1), L.o_spath = take(depth), L.o_rpk = take(2);
  L.o_rbal = take(1), L.o_rnonce = take(1), L.o_rpath = take(depth), L.o_iroot = take(1), L.o_ipath = take(depth);
  return L;
}
// One transaction (or one of its TX_PARTS parts): inputs = the batch's n_public - 1 input signals (standard form), w = its
// slice of K private signals (Montgomery values out), ws = TX_WS_ELEMS scratch elements (parts 0 and 1).  Returns the first
// violated statement of what it ran (TxStmt: the codes are in program order, so the smallest non-zero code over the parts
// is the transaction's); *root = its new root (part 2).
ZKR_HDF uint32_t tx_witness(const Fr *inputs, uint32_t batch, uint32_t depth, uint32_t i, uint32_t K, const TxConsts &k, Fr *w, Fr *ws, Fr *root, int part = -1) {
  const TxLayout L = tx_layout(batch, depth);
  const Fr *in = inputs - 2;  // in[s] = signal s of this batch (s >= 2)
  TxB b;
  b.w = w;
  b.n = 0;
  b.cap = K;
  b.err = ST_OK;
  if (part == 1) ws += 4 * WS_PTS;  // parts 0 and 1 both run scalar-multiplication chains, possibly at the same time
  b.wx = ws, b.wy = ws + WS_PTS, b.wz = ws + 2 * WS_PTS, b.wp = ws + 3 * WS_PTS;
  b.k = k;
  TxInD x;
  bool ok = tx_read_input(in + L.o_root + i, x.root);
  for (int j = 0; j < 8; j++) ok = tx_read_input(in + L.o_tx + 8 * i + j, x.tx[j]) && ok;
  for (int j = 0; j < 2; j++) {
    ok = tx_read_input(in + L.o_spk + 2 * i + j, x.spk[j]) && ok;
    ok = tx_read_input(in + L.o_rpk + 2 * i + j, x.rpk[j]) && ok;
  }
  ok = tx_read_input(in + L.o_sbal + i, x.sbal) && ok;
  ok = tx_read_input(in + L.o_snonce + i, x.snonce) && ok;
  ok = tx_read_input(in + L.o_rbal + i, x.rbal) && ok;
  ok = tx_read_input(in + L.o_rnonce + i, x.rnonce) && ok;
  ok = tx_read_input(in + L.o_iroot + i, x.iroot) && ok;
  for (uint32_t j = 0; j < depth; j++) {
    ok = tx_read_input(in + L.o_spath + depth * i + j, x.spath[j]) && ok;
    ok = tx_read_input(in + L.o_rpath + depth * i + j, x.rpath[j]) && ok;
    ok = tx_read_input(in + L.o_ipath + depth * i + j, x.ipath[j]) && ok;
  }
  if (!ok) b.fail(ST_INPUT_RANGE);
  *root = tx_process(b, x, depth, part, K);
  return b.err;
}
}  // namespace zkr
