// 256-bit special-form prime fields (secp256k1: p = 2^256 - 2^32 - 977, ed25519: p = 2^255 - 19)
// in radix 2^29: 9 limbs, plain residues, lazy reduction with compile-time LIMB bounds.
//
// Reproduces the values of the reference's `_Field` ops (src/abstract/modular.ts:940-982) at the
// boundaries (load / store / comparisons).  In between an element of type Fe9<PR, B> holds limbs
// below B*U (U = 2^29 + 2^19) and a value that is only *congruent* to the canonical residue.
//
// Why this form (measured on MI355X, tools/valu_rates.hip): v_mad_u64_u32 issues in ~2x the time
// of a plain 32-bit VALU add, but every carry instruction (v_addc_co_u32 and friends) costs ~1.65x
// as well, and the 8 x 32-bit form (fp.hpp) spends one per partial product plus carry chains in
// every add / sub.  With 29-bit limbs
//   * a column of 9 products of limbs below A*U and B*U (A*B <= 7) fits a 64-bit accumulator:
//     the product is v_mad_u64_u32 only, one shift + mask per column;
//   * 2^261 = C0 + C1*2^29 (mod p) with C0 < 2^15, C1 < 2^10: the high columns fold into the low
//     ones with one or two more multiply-adds per column, interleaved with the carry chain
//     (two running chains, low `c` and high `d`, no second pass);
//   * a + b is 9 adds, a - b is 9 (sub, add-bias) pairs - no carries, no conditional subtraction;
//     the price is a "weak normalisation" (3 plain ops per limb) where a bound would pass 7.
// Bounds are part of the type and every operation checks them with static_assert; operator* and the
// narrowing conversion insert the normalisation themselves when an operand is too loose.
#pragma once
#include <type_traits>

#include "fp.hpp"
#include "fe9_asm_gen.hpp"

namespace ncg {

constexpr uint32_t FE9_MASK = (1u << 29) - 1u;

template <class PR>
struct Fe9Raw {
  uint32_t v[9];
};

// acc += a * b as ONE v_mad_u64_u32 (the compiler otherwise splits the column sums into separate
// chains and joins them with 64-bit adds and shifts: 27 + 9 extra instructions per product).
NCG_DI void fe9_mac(uint64_t& acc, uint32_t a, uint32_t b) {
#ifdef __HIP_DEVICE_COMPILE__
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
#else
  acc += (uint64_t)a * b;
#endif
}
// acc = a * b: the first product of a chain, the addend is the inline constant 0 (saves clearing the pair)
NCG_DI uint64_t fe9_mul64(uint32_t a, uint32_t b) {
#ifdef __HIP_DEVICE_COMPILE__
  uint64_t acc;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b) : "vcc");
  return acc;
#else
  return (uint64_t)a * b;
#endif
}
// acc += a * k for a wave-uniform constant k (kept in an SGPR)
NCG_DI void fe9_mac_k(uint64_t& acc, uint32_t a, uint32_t k) {
#ifdef __HIP_DEVICE_COMPILE__
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc");
#else
  acc += (uint64_t)a * k;
#endif
}
// acc += a (32-bit) through the multiplier (x 1): one instruction, no carry pair
NCG_DI void fe9_acc32(uint64_t& acc, uint32_t a) {
#ifdef __HIP_DEVICE_COMPILE__
  asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(acc) : "v"(a) : "vcc");
#else
  acc += a;
#endif
}

// The fold / carry tail shared by the product and the square: on entry `c` is the low chain after
// column 7 (incl. u7*C1), `dt` the carry out of column 16, t8 the low limb of column 8.
template <class PR>
NCG_DI void fe9_tail(uint32_t (&r)[9], uint32_t (&t)[9], uint64_t c, uint32_t dt, uint32_t t8) {
  constexpr uint32_t C0 = PR::C0, C1 = PR::C1;
  fe9_mac_k(c, dt, C0);
  fe9_acc32(c, t8);
  t[8] = (uint32_t)c & FE9_MASK;
  c >>= 29;
  if (C1) fe9_mac_k(c, dt, C1);
  // c (below 2^41) has weight 2^261: fold once more into limbs 0..3
  const uint32_t clo = (uint32_t)c & FE9_MASK, chi = (uint32_t)(c >> 29);
  uint64_t e = t[0];
  fe9_mac_k(e, clo, C0);
  r[0] = (uint32_t)e & FE9_MASK;
  e >>= 29;
  if (C1) {
    fe9_acc32(e, t[1]);
    fe9_mac_k(e, clo, C1);
    fe9_mac_k(e, chi, C0);
    r[1] = (uint32_t)e & FE9_MASK;
    e >>= 29;
    fe9_acc32(e, t[2]);
    fe9_mac_k(e, chi, C1);
    r[2] = (uint32_t)e & FE9_MASK;
    r[3] = t[3] + (uint32_t)(e >> 29);
  } else {  // C1 = 0: chi = 0 and the carry is a few bits
    const uint32_t e1 = t[1] + (uint32_t)e;
    r[1] = e1 & FE9_MASK;
    r[2] = t[2] + (e1 >> 29);
    r[3] = t[3];
  }
#pragma unroll
  for (int i = 4; i < 9; i++) r[i] = t[i];
}

#ifndef NCG_FE9_ASM_BLOCKS
#define NCG_FE9_ASM_BLOCKS 1
#endif
#define NCG_FE9_ARGS9(x) x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], x[8]

// The multiply-adds of column pair k as ONE asm block (fe9_asm_gen.hpp, tools/gen_fe9_asm.py): adjacent inline-asm
// STATEMENTS that share a register - every multiply-add names VCC - get a wait state from the compiler's hazard
// recogniser (8 030 s_nop against 10 980 multiply-adds in the round-3 secp256k1 ladder); instructions inside one block do
// not.  `fold`: the block starts with c += u * C1 of the previous column (so that this multiply-add is not a statement of
// its own next to the block).
template <int K, bool SQR>
NCG_DI void fe9_blk(uint64_t& c, uint64_t& d, const uint32_t (&a)[9], const uint32_t (&b)[9], bool fold, uint32_t u, uint32_t C1) {
#define NCG_FE9_CASE(k)                                                                               \
  if constexpr (K == k) {                                                                             \
    if constexpr (SQR) {                                                                              \
      if (fold) fe9_blk_sqr_##k##_f(c, d, NCG_FE9_ARGS9(a), NCG_FE9_ARGS9(b), u, C1);                 \
      else fe9_blk_sqr_##k(c, d, NCG_FE9_ARGS9(a), NCG_FE9_ARGS9(b));                                 \
    } else {                                                                                          \
      if (fold) fe9_blk_mul_##k##_f(c, d, NCG_FE9_ARGS9(a), NCG_FE9_ARGS9(b), u, C1);                 \
      else fe9_blk_mul_##k(c, d, NCG_FE9_ARGS9(a), NCG_FE9_ARGS9(b));                                 \
    }                                                                                                 \
  }
  NCG_FE9_CASE(0) NCG_FE9_CASE(1) NCG_FE9_CASE(2) NCG_FE9_CASE(3) NCG_FE9_CASE(4) NCG_FE9_CASE(5) NCG_FE9_CASE(6) NCG_FE9_CASE(7)
#undef NCG_FE9_CASE
}
// columns 0..7 + their folds, shared by product and square: x = a, y = b (product) or x = a, y = 2a (square)
template <class PR, bool SQR, int K = 0>
NCG_DI void fe9_cols(uint64_t& c, uint64_t& d, uint32_t (&t)[9], const uint32_t (&x)[9], const uint32_t (&y)[9], uint32_t u_prev) {
  constexpr uint32_t C0 = PR::C0, C1 = PR::C1;
  if constexpr (K < 8) {
    fe9_blk<K, SQR>(c, d, x, y, K > 0 && C1 != 0, u_prev, C1);
    const uint32_t u = (uint32_t)d & FE9_MASK;
    d >>= 29;
    fe9_mac_k(c, u, C0);
    t[K] = (uint32_t)c & FE9_MASK;
    c >>= 29;
    if constexpr (K == 7) {
      if (C1) fe9_mac_k(c, u, C1);   // the last fold has no following block to ride in
    }
    fe9_cols<PR, SQR, K + 1>(c, d, t, x, y, u);
  }
}

// r = a * b mod p (congruent), limbs of r below 2^29 + 2.  Requires limbs(a) < A*U, limbs(b) < B*U
// with A*B <= 7 (so that 9 products fit 64 bits) - checked by the callers' types.
template <class PR>
NCG_DI void fe9_mul_limbs(uint32_t (&r)[9], const uint32_t (&a)[9], const uint32_t (&b)[9]) {
  constexpr uint32_t C0 = PR::C0, C1 = PR::C1;
#if NCG_FE9_ASM_BLOCKS
  {
    uint64_t c, d;
    fe9_blk_mul_head(c, d, NCG_FE9_ARGS9(a), NCG_FE9_ARGS9(b));
    const uint32_t t8 = (uint32_t)d & FE9_MASK;
    d >>= 29;
    uint32_t t[9];
    fe9_cols<PR, false>(c, d, t, a, b, 0u);
    fe9_tail<PR>(r, t, c, (uint32_t)d, t8);
    return;
  }
#endif
  uint64_t d = fe9_mul64(a[0], b[8]);  // high chain: columns 8, 9, ..., 16
#pragma unroll
  for (int i = 1; i < 9; i++) fe9_mac(d, a[i], b[8 - i]);
  const uint32_t t8 = (uint32_t)d & FE9_MASK;
  d >>= 29;
  uint64_t c = fe9_mul64(a[0], b[0]);  // low chain: columns 0..8 plus the folded high limbs
  uint32_t t[9];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    // the k+1 products of low column k and the 8-k of high column 9+k, alternating between the two
    // accumulators (back-to-back dependent multiply-adds cost a wait state each)
#pragma unroll
    for (int j = 0; j < 9; j++) {
      if (j <= k && (j | k)) fe9_mac(c, a[j], b[k - j]);
      if (j + k + 1 < 9) fe9_mac(d, a[j + k + 1], b[8 - j]);
    }
    const uint32_t u = (uint32_t)d & FE9_MASK;
    d >>= 29;
    fe9_mac_k(c, u, C0);
    t[k] = (uint32_t)c & FE9_MASK;
    c >>= 29;
    if (C1) fe9_mac_k(c, u, C1);
  }
  fe9_tail<PR>(r, t, c, (uint32_t)d, t8);
}

// r = a^2: off-diagonal products once against the doubled operand (limbs(a) < 2^31 needed).
template <class PR>
NCG_DI void fe9_sqr_limbs(uint32_t (&r)[9], const uint32_t (&a)[9]) {
  constexpr uint32_t C0 = PR::C0, C1 = PR::C1;
  uint32_t a2[9];
#pragma unroll
  for (int i = 0; i < 9; i++) a2[i] = a[i] << 1;
#if NCG_FE9_ASM_BLOCKS
  {
    uint64_t c, d;
    fe9_blk_sqr_head(c, d, NCG_FE9_ARGS9(a), NCG_FE9_ARGS9(a2));
    const uint32_t t8 = (uint32_t)d & FE9_MASK;
    d >>= 29;
    uint32_t t[9];
    fe9_cols<PR, true>(c, d, t, a, a2, 0u);
    fe9_tail<PR>(r, t, c, (uint32_t)d, t8);
    return;
  }
#endif
  // column k = sum_{i<j, i+j=k} 2 a_i a_j + [k even] a_{k/2}^2
  uint64_t d = fe9_mul64(a[4], a[4]);
#pragma unroll
  for (int i = 0; i < 4; i++) fe9_mac(d, a2[i], a[8 - i]);
  const uint32_t t8 = (uint32_t)d & FE9_MASK;
  d >>= 29;
  uint64_t c = fe9_mul64(a[0], a[0]);
  uint32_t t[9];
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) fe9_mac(c, a2[i], a[k - i]);
    if ((k & 1) == 0 && k) fe9_mac(c, a[k / 2], a[k / 2]);
    const int kk = 9 + k;
#pragma unroll
    for (int i = k + 1; 2 * i < kk; i++) fe9_mac(d, a2[i], a[kk - i]);
    if ((kk & 1) == 0) fe9_mac(d, a[kk / 2], a[kk / 2]);
    const uint32_t u = (uint32_t)d & FE9_MASK;
    d >>= 29;
    fe9_mac_k(c, u, C0);
    t[k] = (uint32_t)c & FE9_MASK;
    c >>= 29;
    if (C1) fe9_mac_k(c, u, C1);
  }
  fe9_tail<PR>(r, t, c, (uint32_t)d, t8);
}

// Out-of-line entry points.  The limbs travel as SCALAR arguments: an aggregate argument beyond the
// first is passed by reference through scratch memory by the AMDGPU calling convention (9 stores + 9
// loads + their latency per multiply), scalars go in v0..v17.
template <class PR, class... T>
NCG_MULFN Fe9Raw<PR> fe9_mul_raw_s(T... limbs) {
  static_assert(sizeof...(T) == 18, "two operands of 9 limbs");
  const uint32_t v[18] = {limbs...};
  uint32_t a[9], b[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    a[i] = v[i];
    b[i] = v[9 + i];
  }
  Fe9Raw<PR> r;
  fe9_mul_limbs<PR>(r.v, a, b);
  return r;
}
template <class PR, class... T>
NCG_MULFN Fe9Raw<PR> fe9_sqr_raw_s(T... limbs) {
  static_assert(sizeof...(T) == 9, "one operand of 9 limbs");
  const uint32_t a[9] = {limbs...};
  Fe9Raw<PR> r;
  fe9_sqr_limbs<PR>(r.v, a);
  return r;
}
template <class PR>
NCG_DI Fe9Raw<PR> fe9_mul_raw(const Fe9Raw<PR>& a, const Fe9Raw<PR>& b) {
  return fe9_mul_raw_s<PR>(a.v[0], a.v[1], a.v[2], a.v[3], a.v[4], a.v[5], a.v[6], a.v[7], a.v[8], b.v[0], b.v[1],
                           b.v[2], b.v[3], b.v[4], b.v[5], b.v[6], b.v[7], b.v[8]);
}
template <class PR>
NCG_DI Fe9Raw<PR> fe9_sqr_raw(const Fe9Raw<PR>& a) {
  return fe9_sqr_raw_s<PR>(a.v[0], a.v[1], a.v[2], a.v[3], a.v[4], a.v[5], a.v[6], a.v[7], a.v[8]);
}

template <class PR, int B>
struct Fe9 {
  static_assert(B >= 1 && B <= 7, "Fe9 limb bound out of range (limbs must stay below 2^32)");
  static constexpr int BOUND = B;
  using Params = PR;
  uint32_t v[9];

  Fe9() = default;
  // widening is free; narrowing runs the weak normalisation (result bound 1)
  template <int B2, class = typename std::enable_if<(B2 != B)>::type>
  NCG_DI Fe9(const Fe9<PR, B2>& o) {
    if constexpr (B2 < B) {
#pragma unroll
      for (int i = 0; i < 9; i++) v[i] = o.v[i];
    } else {
      const uint32_t h = o.v[8] >> 29;  // below 8
      v[0] = (o.v[0] & FE9_MASK) + h * PR::C0;
      v[1] = (o.v[1] & FE9_MASK) + (o.v[0] >> 29) + h * PR::C1;
#pragma unroll
      for (int i = 2; i < 9; i++) v[i] = (o.v[i] & FE9_MASK) + (o.v[i - 1] >> 29);
    }
  }
  static NCG_DI Fe9 zero() {
    Fe9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
  }
  static NCG_DI Fe9 one() {
    Fe9 r = zero();
    r.v[0] = 1;
    return r;
  }
  template <class ARR>
  static NCG_DI Fe9 from_limbs(const ARR& a) {  // canonical 29-bit limbs (constants)
    Fe9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a[i];
    return r;
  }
  // literal zero (all limbs): the encoding of "infinity" coordinates; NOT a test mod p
  NCG_DI bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= v[i];
    return o == 0;
  }
};

template <class PR, int B>
NCG_DI Fe9<PR, 1> fe9_norm(const Fe9<PR, B>& a) {
  if constexpr (B == 1) return a;
  else return Fe9<PR, 1>(a);
}

template <class PR, int A, int B>
NCG_DI Fe9<PR, A + B> operator+(const Fe9<PR, A>& a, const Fe9<PR, B>& b) {
  Fe9<PR, A + B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
// a - b = a + BIAS[B] - b, BIAS[B] a multiple of p with every limb in [B*U, (B+1)*U): one v_sad_u32 per limb
// (|BIAS - b| + a, the bias in an SGPR; BIAS >= b by the limb bound) instead of a subtract and an add
template <class PR, int A, int B>
NCG_DI Fe9<PR, A + B + 1> operator-(const Fe9<PR, A>& a, const Fe9<PR, B>& b) {
  Fe9<PR, A + B + 1> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
#ifdef __HIP_DEVICE_COMPILE__
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r.v[i]) : "s"(PR::BIAS[B][i]), "v"(b.v[i]), "v"(a.v[i]));
#else
    r.v[i] = a.v[i] + (PR::BIAS[B][i] - b.v[i]);
#endif
  }
  return r;
}
template <class PR, int A>
NCG_DI Fe9<PR, A + 1> f_neg(const Fe9<PR, A>& a) {
  Fe9<PR, A + 1> r;
  const bool z = a.is_zero();  // keep literal zero literal (the identity's coordinates): -0 = 0
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = z ? 0u : PR::BIAS[A][i] - a.v[i];
  return r;
}
template <class PR, int A>
NCG_DI Fe9<PR, 2 * A> f_dbl(const Fe9<PR, A>& a) {
  Fe9<PR, 2 * A> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] << 1;
  return r;
}

template <class PR, int A, int B>
NCG_DI Fe9<PR, 1> operator*(const Fe9<PR, A>& a, const Fe9<PR, B>& b) {
  if constexpr (A * B > 7) {  // too loose for the 64-bit columns: tighten the looser operand first
    if constexpr (A >= B) return fe9_norm(a) * b;
    else return a * fe9_norm(b);
  } else {
    Fe9Raw<PR> x, y;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      x.v[i] = a.v[i];
      y.v[i] = b.v[i];
    }
    const Fe9Raw<PR> z = fe9_mul_raw<PR>(x, y);
    Fe9<PR, 1> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = z.v[i];
    return r;
  }
}
template <class PR, int A>
NCG_DI Fe9<PR, 1> f_sqr(const Fe9<PR, A>& a) {
  if constexpr (A > 2) {
    return f_sqr(fe9_norm(a));
  } else {
    Fe9Raw<PR> x;
#pragma unroll
    for (int i = 0; i < 9; i++) x.v[i] = a.v[i];
    const Fe9Raw<PR> z = fe9_sqr_raw<PR>(x);
    Fe9<PR, 1> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = z.v[i];
    return r;
  }
}
template <class PR>
NCG_DI Fe9<PR, 1> fe9_sqr_n(Fe9<PR, 1> a, int n) {
  for (int i = 0; i < n; i++) a = f_sqr(a);
  return a;
}

// canonical residue in [0, p) as 9 x 29-bit limbs: sequential carries, two folds of the bits at and
// above the width of p, then conditional subtractions.  Boundary / rare-path code.
template <class PR, int A>
NCG_DI void fe9_canon_limbs(uint32_t (&o)[9], const Fe9<PR, A>& a) {
  // value below 7*U*2^232*(1+2^-29) < 2^264.  Write v = lo + hi * 2^261 and fold: twice.
  uint64_t cy = 0;
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    cy += a.v[i];
    t[i] = (uint32_t)cy & FE9_MASK;
    cy >>= 29;
  }
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {  // cy (below 2^4, then 0 or 1) has weight 2^261
    uint64_t e = (uint64_t)(uint32_t)cy * PR::C0 + t[0];
    t[0] = (uint32_t)e & FE9_MASK;
    e >>= 29;
    e += (uint64_t)(uint32_t)cy * PR::C1 + t[1];
    t[1] = (uint32_t)e & FE9_MASK;
    e >>= 29;
#pragma unroll
    for (int i = 2; i < 9; i++) {
      e += t[i];
      t[i] = (uint32_t)e & FE9_MASK;
      e >>= 29;
    }
    cy = e;
  }
  // now the value is below 2^261: subtract p while >= p.  2^261 / p <= 64 (ed25519), so take the
  // quotient estimate from the top bits first: q = value >> PBITS, value -= q * p  (q * p = q * 2^PBITS - q * c)
  constexpr int PBITS = (PR::P[8] >> 23) ? 256 : 255;  // secp256k1: 256, ed25519: 255
  constexpr int TOPB = PBITS - 232;                    // bits of limb 8 below 2^PBITS
  constexpr uint32_t CSM = (uint32_t)((1ull << 29) - PR::P[0]);  // p = 2^PBITS - c, c = CSM + CS1*2^29
  constexpr uint32_t CS1 = FE9_MASK - PR::P[1];
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const uint32_t q = t[8] >> TOPB;
    t[8] &= (1u << TOPB) - 1u;
    uint64_t e = (uint64_t)q * CSM + t[0];
    t[0] = (uint32_t)e & FE9_MASK;
    e >>= 29;
    e += (uint64_t)q * CS1 + t[1];
    t[1] = (uint32_t)e & FE9_MASK;
    e >>= 29;
#pragma unroll
    for (int i = 2; i < 9; i++) {
      e += t[i];
      t[i] = (uint32_t)e & FE9_MASK;
      e >>= 29;
    }
  }
  // value below 2^PBITS + small: one conditional subtraction of p
  uint32_t s[9];
  int32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    int32_t dd = (int32_t)t[i] - (int32_t)PR::P[i] + bw;
    s[i] = (uint32_t)dd & FE9_MASK;
    bw = dd >> 29;
  }
  const bool ge = bw == 0;
#pragma unroll
  for (int i = 0; i < 9; i++) o[i] = ge ? s[i] : t[i];
}

// a == 0 (mod p).  a = j*p for some j < JMAX*A; the low limb gives j = a0 * p^-1 mod 2^29 - almost
// always j is out of range and the test ends after one multiply.
template <class PR, class... T>
__host__ __device__ __noinline__ bool fe9_eqz_slow(T... limbs) {  // exact test, out of line: rarely reached
  Fe9<PR, 7> a;
  const uint32_t v[9] = {limbs...};
#pragma unroll
  for (int i = 0; i < 9; i++) a.v[i] = v[i];
  uint32_t c[9];
  fe9_canon_limbs<PR, 7>(c, a);
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) o |= c[i];
  return o == 0;
}
template <class PR, int A>
NCG_DI bool f_eqz(const Fe9<PR, A>& a) {
  const uint32_t j = (a.v[0] * PR::PINV) & FE9_MASK;
  if (j >= (uint32_t)(PR::JMAX * A)) return false;
  return fe9_eqz_slow<PR>(a.v[0], a.v[1], a.v[2], a.v[3], a.v[4], a.v[5], a.v[6], a.v[7], a.v[8]);
}
template <class PR, int A, int B>
NCG_DI bool f_eq(const Fe9<PR, A>& a, const Fe9<PR, B>& b) {
  return f_eqz(a - b);
}

// ---- wire format (8 x 32-bit LE limbs, canonical residue) <-> Fe9
template <class PR>
NCG_DI Fe9<PR, 1> fe9_from_wire(const uint32_t* __restrict__ p) {
  uint32_t w[10];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = p[i];
  w[8] = 0;
  w[9] = 0;
  Fe9<PR, 1> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, limb = bit >> 5, sh = bit & 31;
    const uint64_t two = ((uint64_t)w[limb + 1] << 32) | w[limb];
    r.v[i] = (uint32_t)(two >> sh) & FE9_MASK;
  }
  return r;
}
template <class PR, int A>
NCG_DI void fe9_to_wire(uint32_t* __restrict__ p, const Fe9<PR, A>& a) {
  uint32_t l[11];
  uint32_t c[9];
  fe9_canon_limbs<PR, A>(c, a);
#pragma unroll
  for (int i = 0; i < 9; i++) l[i] = c[i];
  l[9] = 0;
  l[10] = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int bit = 32 * k, limb = bit / 29, sh = bit % 29;
    uint64_t acc = (uint64_t)l[limb] >> sh;
    acc |= (uint64_t)l[limb + 1] << (29 - sh);
    acc |= (uint64_t)l[limb + 2] << (58 - sh);
    p[k] = (uint32_t)acc;
  }
}

// Fermat inversion by addition chains (value of modular.ts:159-182 invert; 0 -> 0).
//   secp256k1: p - 2 = 2^256 - 2^32 - 979: 255 squarings + 15 multiplications
//   ed25519:   p - 2 = 2^255 - 21:         254 squarings + 11 multiplications
template <class PR, int A>
NCG_DI Fe9<PR, 1> f_inv(const Fe9<PR, A>& a_in) {
  using F = Fe9<PR, 1>;
  const F x = fe9_norm(a_in);
  if constexpr (PR::C1 != 0) {  // secp256k1
    F x2 = f_sqr(x) * x;
    F x3 = f_sqr(x2) * x;
    F x6 = fe9_sqr_n(x3, 3) * x3;
    F x9 = fe9_sqr_n(x6, 3) * x3;
    F x11 = fe9_sqr_n(x9, 2) * x2;
    F x22 = fe9_sqr_n(x11, 11) * x11;
    F x44 = fe9_sqr_n(x22, 22) * x22;
    F x88 = fe9_sqr_n(x44, 44) * x44;
    F x176 = fe9_sqr_n(x88, 88) * x88;
    F x220 = fe9_sqr_n(x176, 44) * x44;
    F x223 = fe9_sqr_n(x220, 3) * x3;
    F t = fe9_sqr_n(x223, 23) * x22;
    t = fe9_sqr_n(t, 5) * x;
    t = fe9_sqr_n(t, 3) * x2;
    t = fe9_sqr_n(t, 2) * x;
    return t;
  } else {  // ed25519: x^(2^255 - 21) = (x^(2^250 - 1))^(2^5) * x^11
    F z2 = f_sqr(x);
    F z9 = fe9_sqr_n(z2, 2) * x;
    F z11 = z9 * z2;
    F z2_5_0 = f_sqr(z11) * z9;
    F z2_10_0 = fe9_sqr_n(z2_5_0, 5) * z2_5_0;
    F z2_20_0 = fe9_sqr_n(z2_10_0, 10) * z2_10_0;
    F z2_40_0 = fe9_sqr_n(z2_20_0, 20) * z2_20_0;
    F z2_50_0 = fe9_sqr_n(z2_40_0, 10) * z2_10_0;
    F z2_100_0 = fe9_sqr_n(z2_50_0, 50) * z2_50_0;
    F z2_200_0 = fe9_sqr_n(z2_100_0, 100) * z2_100_0;
    F z2_250_0 = fe9_sqr_n(z2_200_0, 50) * z2_50_0;
    return fe9_sqr_n(z2_250_0, 5) * z11;
  }
}

// storage types used by the curve templates: coordinates are kept with limbs below 2*U
using FeSecp = Fe9<Fe9SecpPR, 2>;
using FeEd = Fe9<Fe9EdPR, 2>;

}  // namespace ncg
