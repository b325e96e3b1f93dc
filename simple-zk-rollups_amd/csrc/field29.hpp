// field29.hpp -- BN254 Fq / Fr arithmetic on 9 x 29-bit limbs, Montgomery radix R = 2^261: the representation of the
// DEVICE HOT PATHS (bucket accumulation and reduction of the five MSMs, NTT butterflies; call site of the path:
// /root/reference/operator/src/snarks/common.ts:29).  field.hpp (8 x 32-bit limbs, radix 2^256 -- the radix of the key
// material, /root/reference/operator/src/utils/binarify.ts:78-90) stays the representation of everything that crosses the
// boundary, of the host code and of the cold kernels.
//
// Why.  On gfx950 v_mad_u64_u32 and v_addc_co_u32 BOTH issue at 4 cycles per wave64 (tools/valu_clock.hip, clock read on
// the device: profiles/r2_valu_clock.txt).  With full 32-bit limbs every one of the 136 multiply-adds of a product needs
// a carry word: 272 full-rate instructions, half of them carries.  With 29-bit limbs a column of up to 45 products
// of 58 bits sums in a plain 64-bit accumulator, so a product is 162 multiply-adds and NO carry instructions (about 212
// instructions with the column shifts): 1.4x fewer issue cycles, and the compiler spreads the columns over independent
// accumulators, which removes the dependent-chain stalls of the two-wavefronts-per-SIMD accumulation kernels as well.
// Additions and subtractions become 9 independent 32-bit operations plus one carry sweep; there is no conditional
// subtraction anywhere: R / p = 169, so values are kept LAZILY reduced and the bound of every value is part of its type.
//
// L29<PM, H>: limbs normalised (v[i] < 2^29 for i < 8), value < H * p / 2.  A product accepts operands with
// Ha * Hb <= 676 (a b / R < p) and returns H = 3 or 4; a subtraction adds the smallest multiple of p that keeps it
// non-negative and says so in its result type.  static_asserts check every bound at compile time.
// Plain C++ (no inline assembly): the same code runs on the host for the CPU unit tests (tests/test_host_arith.py).
#pragma once
#include <utility>
#include "field.hpp"
#include "field29_consts.hpp"

namespace zkr {

constexpr uint32_t M29 = (1u << 29) - 1u;

template <class PM, int H>
struct L29 {
  static_assert(H >= 1 && H <= 128, "value bound out of range: the top limb has to stay below 2^28 (reductions take it as a signed word)");
  static constexpr int bound = H;
  uint32_t v[9];
  static ZKR_HD L29 zero() {
    L29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
  }
  // every limb zero: THE representation of the point at infinity's ZZ (a value that is only congruent to zero is not this)
  ZKR_HD bool all_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= v[i];
    return o == 0;
  }
  // a looser bound is always true
  template <int H2>
  ZKR_HD L29<PM, H2> to() const {
    static_assert(H2 >= H, "a value bound can only be relaxed");
    L29<PM, H2> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = v[i];
    return r;
  }
};

// U29<PM, H>: a value below H p / 2 whose limbs are NOT normalised -- each of the low eight is non-negative and below
// W * 2^29 (W = 2: K p - b; W = 3: a + K p - b, limb by limb against a multiple of p whose low limbs were made >= 2^29 - 1 by
// borrowing from the limb above) -- i.e. a subtraction WITHOUT its carry sweep (4 of its 5 instructions per limb).  Legal only
// as one factor of a product whose other factor is normalised: the products of a column then stay below 2^64 as long as
// 9 * (1 + sum over the form's products of Wa Wb) <= 64 (f29_column_fits; 1 = the reduction's own m_i P_j terms), which every
// product form checks at compile time.  The square (doubled limbs) and the additions take normalised operands only.
template <class PM, int H, int W>
struct U29 {
  static_assert(H >= 1 && H <= 128 && (W == 2 || W == 3), "loose operand out of range");
  static constexpr int bound = H;
  uint32_t v[9];
};
// limb weights in units of 2^29 (normalised: 1) and the column condition of a product form
template <class T> struct f29_w;
template <class PM, int H> struct f29_w<L29<PM, H>> { static constexpr int w = 1; using pm = PM; };
template <class PM, int H, int W> struct f29_w<U29<PM, H, W>> { static constexpr int w = W; using pm = PM; };
constexpr bool f29_column_fits(int sum_of_weight_products) { return 9 * (1 + sum_of_weight_products) <= 64; }
// limb i of K p with the low limbs lifted to [2^29 - 1, 2^30): the minuend of the sweep-less subtractions
template <class PM, int K>
constexpr uint32_t f29_lifted(int i) {
  return PM::KP[K][i] + (i < 8 ? (1u << 29) : 0u) - (i > 0 ? 1u : 0u);
}

template <class PM>
ZKR_HD L29<PM, 2> one29() {  // Montgomery 1
  L29<PM, 2> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = PM::ONE[i];
  return r;
}
template <class PM>
ZKR_HD L29<PM, 2> const29(const uint32_t (&c)[9]) {
  L29<PM, 2> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = c[i];
  return r;
}

// ---- 256-bit words <-> limbs.  pack needs value < 2^256 (5.29 p): H <= 10.
template <class PM, int H>
ZKR_HD void pack29(const L29<PM, H> &a, uint32_t (&w)[8]) {
  static_assert(H <= 10, "value may not fit 256 bits: reduce it first (weak())");
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int lo = 32 * k, i = lo / 29, off = lo % 29;
    uint64_t t = (uint64_t)(a.v[i] >> off);
    if (i + 1 <= 8) t |= (uint64_t)a.v[i + 1] << (29 - off);
    if (i + 2 <= 8 && 58 - off < 32) t |= (uint64_t)a.v[i + 2] << (58 - off);
    w[k] = (uint32_t)t;
  }
}
template <class PM, int H>
ZKR_HD L29<PM, H> unpack29(const uint32_t (&w)[8]) {  // the caller states the bound the stored value satisfies
  L29<PM, H> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int lo = 29 * i, k = lo / 32, off = lo % 32;
    uint64_t t = (uint64_t)w[k];
    if (k + 1 <= 7) t |= (uint64_t)w[k + 1] << 32;
    r.v[i] = (uint32_t)(t >> off) & M29;
  }
  return r;
}

// ---- Montgomery reduction shared by every product form.  Column k of the 17: acc += (the form's products of column k)
// + sum_i m_i P_{k-i}; the low 29 bits of the first nine columns are zeroed by m_k = acc * (-p^-1) mod 2^29; then the
// accumulator moves on by 29 bits.  A form is a policy with `column<K>(acc)`; its products of one column must stay below
// 2^63.5 - 9 * 2^58.  The recursion below is the definition and the HOST build (CPU unit tests).  On the device each form
// is ONE inline-assembly statement with the same instruction sequence written out (field29_asm.hpp, generated): 162
// multiply-adds + 17 shifts + 17 masks + 9 m_k products = 205 instructions for x * y.  Assembly not for the instructions
// (the compiler emits the same v_mad_u64_u32 from `acc += (uint64_t)x * y`) but because (a) it re-associates a column's sum
// into partial chains, one extra 64-bit addition per column, and (b) on gfx950 it pads the first reader of every register
// an asm statement wrote with `s_nop 0`: with one statement per column chain (the first device form of this file) that was
// ~50 idle issue cycles per product.  Measured, G Fq-mul/s at 2 wavefronts per SIMD: 159.8 plain C++, 168.8 chains as asm
// statements (tools/mul29_test.hip); the whole product as one statement: +2.0 % proofs/s at 2^20 in a same-box A/B of
// the two builds (profiles/r2_07_ab_product_asm.txt).  A dependent chain of v_mad_u64_u32 issues as fast as independent
// ones (tools/dep_chain.hip: 4.76 / 4.59 / 4.56 cycles per instruction for 1 / 2 / 4 chains at 2 wavefronts per SIMD), so
// nothing is lost by the serial order inside a statement.
template <int LO, int... I>
constexpr auto f29_range_seq(std::integer_sequence<int, I...>) { return std::integer_sequence<int, (LO + I)...>{}; }
template <int LO, int HI>
using f29_range = decltype(f29_range_seq<LO>(std::make_integer_sequence<int, (HI >= LO ? HI - LO + 1 : 0)>{}));
constexpr int f29_max(int a, int b) { return a > b ? a : b; }
constexpr int f29_min(int a, int b) { return a < b ? a : b; }

#if defined(__HIP_DEVICE_COMPILE__)
#include "field29_asm.hpp"
#define ZKR_F29_DEVICE_ASM 1
#else
#define ZKR_F29_DEVICE_ASM 0
#endif

// acc += sum_{i in I} x[i] * y[K - i]
template <int K, int... I>
ZKR_HD void f29_cross(uint64_t &acc, const uint32_t (&x)[9], const uint32_t (&y)[9], std::integer_sequence<int, I...>) {
  if constexpr (sizeof...(I) > 0) {
    ((acc += (uint64_t)x[I] * y[K - I]), ...);
  }
}
// acc += sum_{i in I} m[i] * P[K - i]
template <class PM, int K, int... I>
ZKR_HD void f29_reduce_terms(uint64_t &acc, const uint32_t (&m)[9], std::integer_sequence<int, I...>) {
  if constexpr (sizeof...(I) > 0) {
    ((acc += (uint64_t)m[I] * PM::P[K - I]), ...);
  }
}
template <class PM, int K, class Cols>
ZKR_HD void mont29_from(uint32_t (&out)[9], uint32_t (&m)[9], uint64_t &acc, const Cols &cols) {
  cols.template column<K>(acc);
  f29_reduce_terms<PM, K>(acc, m, f29_range<f29_max(0, K - 8), f29_min(K - 1, 8)>{});
  if constexpr (K < 9) {
    m[K] = ((uint32_t)acc * PM::INV) & M29;
    f29_reduce_terms<PM, K>(acc, m, std::integer_sequence<int, K>{});  // + m_K * P_0
    acc >>= 29;
  } else {
    out[K - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  if constexpr (K < 16) mont29_from<PM, K + 1>(out, m, acc, cols);
  else out[8] = (uint32_t)acc;
}
template <class PM, class Cols>
ZKR_HD void mont29(uint32_t (&out)[9], const Cols &cols) {
  uint32_t m[9];
  uint64_t acc = 0;
  mont29_from<PM, 0>(out, m, acc, cols);
}

// the product forms: x * y; the square (off-diagonal products once, against the doubled limbs); sums of 2 and of 4 products
struct F29Mul {
  const uint32_t (&x)[9], (&y)[9];
  template <int K> ZKR_HD void column(uint64_t &acc) const { f29_cross<K>(acc, x, y, f29_range<f29_max(0, K - 8), f29_min(K, 8)>{}); }
};
struct F29Sqr {
  const uint32_t (&x)[9], (&d)[9];  // d = 2 x
  template <int K> ZKR_HD void column(uint64_t &acc) const {
    f29_cross<K>(acc, x, d, f29_range<f29_max(0, K - 8), (K + 1) / 2 - 1>{});      // i < j = K - i
    if constexpr (K % 2 == 0) f29_cross<K>(acc, x, x, std::integer_sequence<int, K / 2>{});
  }
};
struct F29Sum2 {
  const uint32_t (&a)[9], (&b)[9], (&c)[9], (&d)[9];
  template <int K> ZKR_HD void column(uint64_t &acc) const {
    using R = f29_range<f29_max(0, K - 8), f29_min(K, 8)>;
    f29_cross<K>(acc, a, b, R{});
    f29_cross<K>(acc, c, d, R{});
  }
};
struct F29Sum4 {
  const uint32_t (&a)[9], (&b)[9], (&c)[9], (&d)[9], (&e)[9], (&f)[9], (&g)[9], (&h)[9];
  template <int K> ZKR_HD void column(uint64_t &acc) const {
    using R = f29_range<f29_max(0, K - 8), f29_min(K, 8)>;
    f29_cross<K>(acc, a, b, R{});
    f29_cross<K>(acc, c, d, R{});
    f29_cross<K>(acc, e, f, R{});
    f29_cross<K>(acc, g, h, R{});
  }
};

constexpr int mul_out_h(int q) { return q <= 338 ? 3 : 4; }  // a b / R + p with a b <= q p^2 / 4 and R / p = 169.28

template <class PM, int HA, int HB>
ZKR_HD L29<PM, mul_out_h(HA * HB)> mul(const L29<PM, HA> &a, const L29<PM, HB> &b) {
  static_assert(HA * HB <= 676, "product of the operand bounds exceeds R / p");
  L29<PM, mul_out_h(HA * HB)> r;
#if ZKR_F29_DEVICE_ASM
  mont29_asm_mul<PM>(r.v, a.v, b.v);
#else
  mont29<PM>(r.v, F29Mul{a.v, b.v});
#endif
  return r;
}
// one factor without its carry sweep (U29)
template <class PM, int HA, int HB, int WB>
ZKR_HD L29<PM, mul_out_h(HA * HB)> mul(const L29<PM, HA> &a, const U29<PM, HB, WB> &b) {
  static_assert(HA * HB <= 676, "product of the operand bounds exceeds R / p");
  static_assert(f29_column_fits(WB), "column sum of the product exceeds 64 bits");
  L29<PM, mul_out_h(HA * HB)> r;
#if ZKR_F29_DEVICE_ASM
  mont29_asm_mul<PM>(r.v, a.v, b.v);
#else
  mont29<PM>(r.v, F29Mul{a.v, b.v});
#endif
  return r;
}

// a^2: the off-diagonal products are taken once against the doubled limbs (2 a_j < 2^30 still multiplies without
// overflow): 45 multiply-adds instead of 81 in front of the same reduction
template <class PM, int HA>
ZKR_HD L29<PM, mul_out_h(HA * HA)> sqr(const L29<PM, HA> &a) {
  static_assert(HA * HA <= 676, "square of the operand bound exceeds R / p");
  uint32_t d[9];
#pragma unroll
  for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
  L29<PM, mul_out_h(HA * HA)> r;
#if ZKR_F29_DEVICE_ASM
  mont29_asm_sqr<PM>(r.v, a.v, d);
#else
  mont29<PM>(r.v, F29Sqr{a.v, d});
#endif
  return r;
}

// a b + c d with one reduction.  A, B, C, D: L29 or -- at most one factor of each product -- U29
template <class A, class B, class C, class D>
ZKR_HD L29<typename f29_w<A>::pm, mul_out_h(A::bound * B::bound + C::bound * D::bound)> mul_sum2(const A &a, const B &b, const C &c, const D &d) {
  using PM = typename f29_w<A>::pm;
  static_assert(A::bound * B::bound + C::bound * D::bound <= 676, "sum of the products of the operand bounds exceeds R / p");
  static_assert((f29_w<A>::w == 1 || f29_w<B>::w == 1) && (f29_w<C>::w == 1 || f29_w<D>::w == 1), "one factor of every product has to be normalised");
  static_assert(f29_column_fits(f29_w<A>::w * f29_w<B>::w + f29_w<C>::w * f29_w<D>::w), "column sum of the products exceeds 64 bits");
  L29<PM, mul_out_h(A::bound * B::bound + C::bound * D::bound)> r;
#if ZKR_F29_DEVICE_ASM
  mont29_asm_sum2<PM>(r.v, a.v, b.v, c.v, d.v);
#else
  mont29<PM>(r.v, F29Sum2{a.v, b.v, c.v, d.v});
#endif
  return r;
}
// a b + c d + e f + g h with one reduction (with normalised factors: 36 products of 58 bits and the 9 of the reduction)
template <class A, class B, class C, class D, class E, class F, class G, class I>
ZKR_HD L29<typename f29_w<A>::pm, mul_out_h(A::bound * B::bound + C::bound * D::bound + E::bound * F::bound + G::bound * I::bound)>
mul_sum4(const A &a, const B &b, const C &c, const D &d, const E &e, const F &f, const G &g, const I &h) {
  using PM = typename f29_w<A>::pm;
  constexpr int Q = A::bound * B::bound + C::bound * D::bound + E::bound * F::bound + G::bound * I::bound;
  static_assert(Q <= 676, "sum of the products of the operand bounds exceeds R / p");
  static_assert((f29_w<A>::w == 1 || f29_w<B>::w == 1) && (f29_w<C>::w == 1 || f29_w<D>::w == 1) && (f29_w<E>::w == 1 || f29_w<F>::w == 1) && (f29_w<G>::w == 1 || f29_w<I>::w == 1),
                "one factor of every product has to be normalised");
  static_assert(f29_column_fits(f29_w<A>::w * f29_w<B>::w + f29_w<C>::w * f29_w<D>::w + f29_w<E>::w * f29_w<F>::w + f29_w<G>::w * f29_w<I>::w), "column sum of the products exceeds 64 bits");
  L29<PM, mul_out_h(Q)> r;
#if ZKR_F29_DEVICE_ASM
  mont29_asm_sum4<PM>(r.v, a.v, b.v, c.v, d.v, e.v, f.v, g.v, h.v);
#else
  mont29<PM>(r.v, F29Sum4{a.v, b.v, c.v, d.v, e.v, f.v, g.v, h.v});
#endif
  return r;
}

// the four product forms on raw limbs, as this build runs them (device: field29_asm.hpp; host: the recursion above):
// 0: a b   1: a^2   2: a b + c d   3: a b + c d + e f + g h.  The limb-level self test compares the two builds.
template <class PM>
ZKR_HD void f29_raw_form(int form, const uint32_t (&op)[8][9], uint32_t (&r)[9]) {
  uint32_t d[9];
#pragma unroll
  for (int i = 0; i < 9; i++) d[i] = op[0][i] << 1;
#if ZKR_F29_DEVICE_ASM
  if (form == 0) mont29_asm_mul<PM>(r, op[0], op[1]);
  else if (form == 1) mont29_asm_sqr<PM>(r, op[0], d);
  else if (form == 2) mont29_asm_sum2<PM>(r, op[0], op[1], op[2], op[3]);
  else mont29_asm_sum4<PM>(r, op[0], op[1], op[2], op[3], op[4], op[5], op[6], op[7]);
#else
  if (form == 0) mont29<PM>(r, F29Mul{op[0], op[1]});
  else if (form == 1) mont29<PM>(r, F29Sqr{op[0], d});
  else if (form == 2) mont29<PM>(r, F29Sum2{op[0], op[1], op[2], op[3]});
  else mont29<PM>(r, F29Sum4{op[0], op[1], op[2], op[3], op[4], op[5], op[6], op[7]});
#endif
}

// a + b: limb-wise, then one carry sweep
template <class PM, int HA, int HB>
ZKR_HD L29<PM, HA + HB> add(const L29<PM, HA> &a, const L29<PM, HB> &b) {
  L29<PM, HA + HB> r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint32_t s = a.v[i] + b.v[i] + c;
    if (i < 8) { r.v[i] = s & M29; c = s >> 29; } else r.v[i] = s;
  }
  return r;
}
template <class PM, int HA>
ZKR_HD L29<PM, 2 * HA> dbl(const L29<PM, HA> &a) {
  return add(a, a);
}

// a - b + K p, K = ceil(HB / 2) (the smallest multiple of p that is >= every value b can take): limb-wise with signed
// carries, so no limb of the constant has to dominate the limb it meets
template <class PM, int HA, int HB>
ZKR_HD L29<PM, HA + 2 * ((HB + 1) / 2)> sub(const L29<PM, HA> &a, const L29<PM, HB> &b) {
  constexpr int K = (HB + 1) / 2;
  static_assert(K <= 24, "no multiple of p tabulated for this bound");
  L29<PM, HA + 2 * K> r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    int32_t d = (int32_t)(a.v[i] + PM::KP[K][i]) - (int32_t)b.v[i] + c;
    if (i < 8) { r.v[i] = (uint32_t)d & M29; c = d >> 29; } else r.v[i] = (uint32_t)d;  // the total is >= 0, so the top limb is
  }
  return r;
}
// a - b - 2 c + K p in ONE carry sweep (the X coordinate of every group addition: R^2 - P^3 - 2 X1 P^2), K = ceil((HB + 2 HC) / 2)
template <class PM, int HA, int HB, int HC>
ZKR_HD L29<PM, HA + 2 * ((HB + 2 * HC + 1) / 2)> sub_sub_dbl(const L29<PM, HA> &a, const L29<PM, HB> &b, const L29<PM, HC> &c) {
  constexpr int K = (HB + 2 * HC + 1) / 2;
  static_assert(K <= 24, "no multiple of p tabulated for this bound");
  L29<PM, HA + 2 * K> r;
  int32_t cy = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    int32_t d = (int32_t)(a.v[i] + PM::KP[K][i]) - (int32_t)b.v[i] - (int32_t)(c.v[i] << 1) + cy;  // > -2^31: three limbs below 2^29 subtracted
    if (i < 8) { r.v[i] = (uint32_t)d & M29; cy = d >> 29; } else r.v[i] = (uint32_t)d;
  }
  return r;
}
// K p - b
template <class PM, int HB>
ZKR_HD L29<PM, 2 * ((HB + 1) / 2)> neg(const L29<PM, HB> &b) {
  constexpr int K = (HB + 1) / 2;
  static_assert(K <= 24, "no multiple of p tabulated for this bound");
  L29<PM, 2 * K> r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    int32_t d = (int32_t)PM::KP[K][i] - (int32_t)b.v[i] + c;
    if (i < 8) { r.v[i] = (uint32_t)d & M29; c = d >> 29; } else r.v[i] = (uint32_t)d;
  }
  return r;
}
// ---- the same differences WITHOUT the carry sweep (U29): for values that only ever become one factor of a product.
// K' p - b with K' = ceil(HB / 2) + 1: the extra p keeps the top limb non-negative (b's top limb is at most that of
// ceil(HB / 2) p, and one more p adds P[8] > 1 to it); the low limbs: f29_lifted - b_i in [0, 2^30)
template <class PM, int HB>
ZKR_HD U29<PM, 2 * ((HB + 1) / 2 + 1), 2> neg_loose(const L29<PM, HB> &b) {
  constexpr int K = (HB + 1) / 2 + 1;
  static_assert(K <= 24, "no multiple of p tabulated for this bound");
  U29<PM, 2 * K, 2> r;
  constexpr uint32_t c[9] = {f29_lifted<PM, K>(0), f29_lifted<PM, K>(1), f29_lifted<PM, K>(2), f29_lifted<PM, K>(3), f29_lifted<PM, K>(4),
                             f29_lifted<PM, K>(5), f29_lifted<PM, K>(6), f29_lifted<PM, K>(7), f29_lifted<PM, K>(8)};
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = c[i] - b.v[i];
  return r;
}
// a + K' p - b, limbs in [0, 3 * 2^29)
template <class PM, int HA, int HB>
ZKR_HD U29<PM, HA + 2 * ((HB + 1) / 2 + 1), 3> sub_loose(const L29<PM, HA> &a, const L29<PM, HB> &b) {
  constexpr int K = (HB + 1) / 2 + 1;
  static_assert(K <= 24, "no multiple of p tabulated for this bound");
  U29<PM, HA + 2 * K, 3> r;
  constexpr uint32_t c[9] = {f29_lifted<PM, K>(0), f29_lifted<PM, K>(1), f29_lifted<PM, K>(2), f29_lifted<PM, K>(3), f29_lifted<PM, K>(4),
                             f29_lifted<PM, K>(5), f29_lifted<PM, K>(6), f29_lifted<PM, K>(7), f29_lifted<PM, K>(8)};
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + c[i] - b.v[i];
  return r;
}
// flip ? -b : b as a product factor: the sign of a table point folded into its y (18 instructions instead of a negation
// with its sweep and a select)
template <class PM, int HB>
ZKR_HD U29<PM, 2 * ((HB + 1) / 2 + 1), 2> cneg_loose(const L29<PM, HB> &b, bool flip) {
  auto n = neg_loose(b);
#pragma unroll
  for (int i = 0; i < 9; i++) n.v[i] = flip ? n.v[i] : b.v[i];
  return n;
}
// a b - c d with one reduction; b may come without its sweep
template <class A, class B, class PM, int HC, class D>
ZKR_HD auto mul_sub(const A &a, const B &b, const L29<PM, HC> &c, const D &d) {
  return mul_sum2(a, b, neg_loose(c), d);
}

// (flip ? -y : y) z and a - b as the factor of the Y coordinate's product sum: the forms the group law asks for, per field
template <class PM, int HY, int HZ>
ZKR_HD auto mul_cneg(const L29<PM, HY> &y, bool flip, const L29<PM, HZ> &z) { return mul(z, cneg_loose(y, flip)); }
template <class PM, int HA, int HB>
ZKR_HD auto sub_factor(const L29<PM, HA> &a, const L29<PM, HB> &b) { return sub_loose(a, b); }

// value < 1.5 p whatever it was (one product with the Montgomery 1): before a value with a wide bound is stored
template <class PM, int HA>
ZKR_HD L29<PM, 3> weak(const L29<PM, HA> &a) {
  static_assert(HA * 2 <= 338, "bound too wide even for a product with one");
  return mul(a, one29<PM>()).template to<3>();
}

// value < 2.5 p whatever it was, WITHOUT a product: one quotient estimate from the top limb (the NTT's bound keeper, where a
// product with one would double the cost of a butterfly).  t = v[8] = floor(V / 2^232), D = P[8] + 1 > p / 2^232,
// q = floor(t * floor(2^32 / D) / 2^32) is floor(t / D) or one less, so 0 <= V - q p: q p <= (t / D) p < t 2^232 <= V; and
// V - q p < (t + 1) 2^232 - (t / D - 2) p <= 2 p + 2^232 + (t / D) 2^232 < 2.001 p.  About 60 instructions.
template <class PM, int HA>
ZKR_HD L29<PM, 5> barrett(const L29<PM, HA> &a) {
  constexpr uint32_t D = PM::P[8] + 1;
  constexpr uint32_t RECIP = (uint32_t)((1ull << 32) / D);
  const uint32_t q = (uint32_t)(((uint64_t)a.v[8] * RECIP) >> 32);  // < 2^7 for every bound the type admits
  L29<PM, 5> r;
  uint64_t t = 0;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    t += (uint64_t)q * PM::P[i];
    const uint32_t ti = i < 8 ? (uint32_t)t & M29 : (uint32_t)t;  // limb i of q p
    t >>= 29;
    const int32_t d = (int32_t)a.v[i] - (int32_t)ti + c;
    if (i < 8) { r.v[i] = (uint32_t)d & M29; c = d >> 29; } else r.v[i] = (uint32_t)d;
  }
  return r;
}
// the canonical residue of a value with a small bound, inline (canonical() below is the out-of-line form for cold code)
template <class PM, int HA>
ZKR_HD L29<PM, 2> canonical_small(const L29<PM, HA> &a) {
  static_assert(HA <= 6, "reduce first (barrett / weak): one conditional subtraction per half-modulus pair");
  L29<PM, 2> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i];
#pragma unroll
  for (int it = 0; it < (HA - 1) / 2; it++) {  // value < HA p / 2: after (HA - 1) / 2 conditional subtractions it is below p
    int32_t c = 0;
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int32_t t = (int32_t)r.v[i] - (int32_t)PM::P[i] + c;
      if (i < 8) { d[i] = (uint32_t)t & M29; c = t >> 29; } else { d[i] = (uint32_t)t; c = t >> 31; }
    }
    const bool ge = c == 0;  // no borrow out of the top limb: the value was >= p
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = ge ? d[i] : r.v[i];
  }
  return r;
}

// a == 0 (mod p) for a value below 2 p (every product): it is 0 or p
template <class PM, int HA>
ZKR_HD bool is_zero_mod_p(const L29<PM, HA> &a) {
  static_assert(HA <= 4, "reduce first: the test compares with 0 and p only");
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) { z |= a.v[i]; e |= a.v[i] ^ PM::P[i]; }
  return z == 0 || e == 0;
}
// the cheap half of that test, for the hot loops: a value that is 0 or p has limb 0 equal to 0 or P[0]; everything else is
// sorted out by two compares (one value in 2^28 passes by accident and then takes the full test)
template <class PM, int HA>
ZKR_HD bool maybe_zero_mod_p(const L29<PM, HA> &a) {
  static_assert(HA <= 4, "reduce first: the test compares with 0 and p only");
  return a.v[0] == 0 || a.v[0] == PM::P[0];
}
// the canonical residue (< p), for values that leave the device: at most HA / 2 subtractions of p
template <class PM, int HA>
ZKR_HD_COLD L29<PM, 2> canonical(const L29<PM, HA> &a) {
  L29<PM, 2> r;
  uint32_t v[9];
#pragma unroll
  for (int i = 0; i < 9; i++) v[i] = a.v[i];
  for (int it = 0; it < (HA + 1) / 2; it++) {
    int32_t c = 0;
    uint32_t d[9];
    for (int i = 0; i < 9; i++) {
      int32_t t = (int32_t)v[i] - (int32_t)PM::P[i] + c;
      if (i < 8) { d[i] = (uint32_t)t & M29; c = t >> 29; } else { d[i] = (uint32_t)t; c = t >> 31; }
    }
    if (c == 0)  // no borrow out of the top limb: v >= p
      for (int i = 0; i < 9; i++) v[i] = d[i];
  }
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = v[i];
  return r;
}

// ---- between the two representations (device side of the boundary)
// Fp (8 x 32, any radix, value < 2^256 -> bound 11 half-moduli) as limbs
template <class PM29, class PM, int H = 2>
ZKR_HD L29<PM29, H> limbs_of(const Fp<PM> &a) {
  return unpack29<PM29, H>(a.v);
}
template <class PM, class PM29, int H>
ZKR_HD Fp<PM> words_of(const L29<PM29, H> &a) {
  Fp<PM> r;
  pack29(a, r.v);
  return r;
}

using Fq29 = Fq29Params;
using Fr29 = Fr29Params;

// ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1) over the lazy limbs
template <int H>
struct Q29 {
  L29<Fq29, H> a, b;  // a + b u
  static ZKR_HD Q29 zero() { return Q29{L29<Fq29, H>::zero(), L29<Fq29, H>::zero()}; }
  ZKR_HD bool all_zero() const { return a.all_zero() && b.all_zero(); }
  template <int H2>
  ZKR_HD Q29<H2> to() const { return Q29<H2>{a.template to<H2>(), b.template to<H2>()}; }
};
constexpr int neg_h(int h) { return 2 * ((h + 1) / 2); }
template <int HA, int HB>
ZKR_HD Q29<HA + HB> add(const Q29<HA> &x, const Q29<HB> &y) { return Q29<HA + HB>{add(x.a, y.a), add(x.b, y.b)}; }
template <int HA>
ZKR_HD Q29<2 * HA> dbl(const Q29<HA> &x) { return Q29<2 * HA>{dbl(x.a), dbl(x.b)}; }
template <int HA, int HB>
ZKR_HD Q29<HA + neg_h(HB)> sub(const Q29<HA> &x, const Q29<HB> &y) { return Q29<HA + neg_h(HB)>{sub(x.a, y.a), sub(x.b, y.b)}; }
template <int HB>
ZKR_HD Q29<neg_h(HB)> neg(const Q29<HB> &y) { return Q29<neg_h(HB)>{neg(y.a), neg(y.b)}; }
template <int HA, int HB, int HC>
ZKR_HD Q29<HA + 2 * ((HB + 2 * HC + 1) / 2)> sub_sub_dbl(const Q29<HA> &x, const Q29<HB> &y, const Q29<HC> &z) {
  return Q29<HA + 2 * ((HB + 2 * HC + 1) / 2)>{sub_sub_dbl(x.a, y.a, z.a), sub_sub_dbl(x.b, y.b, z.b)};
}
constexpr int max_h(int a, int b) { return a > b ? a : b; }
// (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u: each component a sum of two products with one reduction
template <int HA, int HB>
ZKR_HD auto mul(const Q29<HA> &x, const Q29<HB> &y) {
  auto re = mul_sum2(x.a, y.a, neg_loose(x.b), y.b);
  auto im = mul_sum2(x.a, y.b, x.b, y.a);
  constexpr int H = max_h(decltype(re)::bound, decltype(im)::bound);
  return Q29<H>{re.template to<H>(), im.template to<H>()};
}
// (a^2 - b^2) + 2 a b u
template <int HA>
ZKR_HD auto sqr(const Q29<HA> &x) {
  auto re = mul_sum2(x.a, x.a, neg_loose(x.b), x.b);
  auto im = mul(dbl(x.a), x.b);
  constexpr int H = max_h(decltype(re)::bound, decltype(im)::bound);
  return Q29<H>{re.template to<H>(), im.template to<H>()};
}
// x y - z w: each component a sum of four products with one reduction
template <int HX, int HY, int HZ, int HW>
ZKR_HD auto mul_sub(const Q29<HX> &x, const Q29<HY> &y, const Q29<HZ> &z, const Q29<HW> &w) {
  auto nza = neg_loose(z.a);  // the negations as product factors: no carry sweeps (U29)
  auto re = mul_sum4(x.a, y.a, neg_loose(x.b), y.b, nza, w.a, z.b, w.b);
  auto im = mul_sum4(x.a, y.b, x.b, y.a, nza, w.b, neg_loose(z.b), w.a);
  constexpr int H = max_h(decltype(re)::bound, decltype(im)::bound);
  return Q29<H>{re.template to<H>(), im.template to<H>()};
}
// (flip ? -y : y) z: both signs of y's components as sweep-less factors, selected per lane
template <int HY, int HZ>
ZKR_HD auto mul_cneg(const Q29<HY> &y, bool flip, const Q29<HZ> &z) {
  auto pa = neg_loose(y.a), pb = neg_loose(y.b), nb = pb;  // pa, pb become +-y.a, +-y.b; nb the negative of pb
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint32_t na = pa.v[i], nbi = pb.v[i];
    pa.v[i] = flip ? na : y.a.v[i];
    pb.v[i] = flip ? nbi : y.b.v[i];
    nb.v[i] = flip ? y.b.v[i] : nbi;
  }
  auto re = mul_sum2(z.a, pa, z.b, nb);
  auto im = mul_sum2(z.b, pa, z.a, pb);
  constexpr int H = max_h(decltype(re)::bound, decltype(im)::bound);
  return Q29<H>{re.template to<H>(), im.template to<H>()};
}
// over Fq2 the Y coordinate is a sum of FOUR products per component: its column budget has no room for a factor of weight 3
template <int HA, int HB>
ZKR_HD auto sub_factor(const Q29<HA> &x, const Q29<HB> &y) { return sub(x, y); }
template <int HA>
ZKR_HD Q29<3> weak(const Q29<HA> &x) { return Q29<3>{weak(x.a), weak(x.b)}; }
template <int HA>
ZKR_HD bool is_zero_mod_p(const Q29<HA> &x) { return is_zero_mod_p(x.a) && is_zero_mod_p(x.b); }
template <int HA>
ZKR_HD bool maybe_zero_mod_p(const Q29<HA> &x) { return maybe_zero_mod_p(x.a); }
template <int HA>
ZKR_HD_COLD Q29<2> canonical(const Q29<HA> &x) { return Q29<2>{canonical(x.a), canonical(x.b)}; }

template <int HA>
ZKR_HD Q29<5> barrett(const Q29<HA> &x) { return Q29<5>{barrett(x.a), barrett(x.b)}; }
template <int HA>
ZKR_HD Q29<2> canonical_small(const Q29<HA> &x) { return Q29<2>{canonical_small(x.a), canonical_small(x.b)}; }

// ---- inversion (key-build kernels only): a^(p-2), plain square-and-multiply over the bits of p - 2 (254 squares and one
// product per set bit; the exponent is the same for every lane, so the branch is uniform).  0 -> 0.
template <class PM, int HA>
ZKR_HD L29<PM, 4> inv29(const L29<PM, HA> &a) {
  static_assert(HA * 4 <= 676, "operand bound too wide for the chain's products");
  static_assert(PM::P[0] >= 2, "p - 2 is taken limb-wise without a borrow");
  L29<PM, 4> r = one29<PM>().template to<4>();
#pragma unroll 1
  for (int i = 253; i >= 0; i--) {  // p < 2^254
    r = sqr(r).template to<4>();
    const uint32_t limb = PM::P[i / 29] - (i / 29 == 0 ? 2u : 0u);
    if ((limb >> (i % 29)) & 1u) r = mul(r, a).template to<4>();
  }
  return r;
}
// 1 / (a + b u) = (a - b u) / (a^2 + b^2)
template <int HA>
ZKR_HD Q29<4> inv29(const Q29<HA> &x) {
  auto ni = inv29(mul_sum2(x.a, x.a, x.b, x.b));
  return Q29<4>{mul(x.a, ni).template to<4>(), mul(neg(x.b), ni).template to<4>()};
}

}  // namespace zkr
