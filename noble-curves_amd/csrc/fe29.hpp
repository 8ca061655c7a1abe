// bls12-381 base field in radix 2^29 (14 limbs, R = 2^406), Montgomery form, lazy reduction
// with compile-time value bounds.
//
// Values of the reference's `_Field` ops (src/abstract/modular.ts:940-982) are reproduced
// exactly at the boundaries (load / store / comparisons); in between an element of type
// Fe29<B> holds limbs < 2^29 and a value < B*p that is only *congruent* to the canonical
// residue.  406 - 381 = 25 spare bits mean a Montgomery product of operands below 2^12 p stays
// below 2p, so additions and subtractions never reduce: `a + b` is Fe29<A+B>, `a - b` adds the
// smallest 2^k p >= B*p first and is Fe29<A + 2^k>, `a * b` is Fe29<2>; every bound is checked
// by static_assert.  The multiply itself (fp29.hpp) is pure v_mad_u64_u32 into 64-bit column
// accumulators - no carry instructions - which is what the integer pipe of gfx950 wants.
#pragma once
#include <type_traits>

#include "fp29.hpp"

namespace ncg {

constexpr int fe29_pow2ceil_log(int b) {
  int k = 0;
  while ((1 << k) < b) k++;
  return k;
}

template <int B>
struct Fe29 {
  static_assert(B >= 1 && B <= 4096, "Fe29 bound out of range");
  static constexpr int N = 14;
  static constexpr int BOUND = B;
  using PR = ParamsBls29;
  uint32_t v[N];

  Fe29() = default;
  template <int B2, class = typename std::enable_if<(B2 < B)>::type>
  NCG_DI Fe29(const Fe29<B2>& o) {  // widening is free
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = o.v[i];
  }
  static NCG_DI Fe29 zero() {
    Fe29 r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = 0;
    return r;
  }
  static NCG_DI Fe29 one() {
    Fe29 r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = PR::R1[i];
    return r;
  }
  // literal zero (all limbs): the encoding of "infinity" coordinates; NOT a test mod p
  NCG_DI bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= v[i];
    return o == 0;
  }
};

template <int A, int B>
NCG_DI Fe29<A + B> operator+(const Fe29<A>& a, const Fe29<B>& b) {
  constexpr uint32_t MASK = (1u << 29) - 1u;
  Fe29<A + B> r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 13; i++) {
    uint32_t t = a.v[i] + b.v[i] + c;
    r.v[i] = t & MASK;
    c = t >> 29;
  }
  r.v[13] = a.v[13] + b.v[13] + c;
  return r;
}

template <int A, int B>
NCG_DI Fe29<A + (1 << fe29_pow2ceil_log(B))> operator-(const Fe29<A>& a, const Fe29<B>& b) {
  constexpr int K = fe29_pow2ceil_log(B);
  constexpr uint32_t MASK = (1u << 29) - 1u;
  Fe29<A + (1 << K)> r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 13; i++) {
    int32_t t = (int32_t)a.v[i] + (int32_t)ParamsBls29::PMUL[K][i] - (int32_t)b.v[i] + c;
    r.v[i] = (uint32_t)t & MASK;
    c = t >> 29;  // arithmetic shift: borrow propagates as -1
  }
  r.v[13] = (uint32_t)((int32_t)a.v[13] + (int32_t)ParamsBls29::PMUL[K][13] - (int32_t)b.v[13] + c);
  return r;
}

template <int A>
NCG_DI Fe29<(1 << fe29_pow2ceil_log(A))> f_neg(const Fe29<A>& a) {  // 2^K p - a
  constexpr int K = fe29_pow2ceil_log(A);
  constexpr uint32_t MASK = (1u << 29) - 1u;
  Fe29<(1 << K)> r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 13; i++) {
    int32_t t = (int32_t)ParamsBls29::PMUL[K][i] - (int32_t)a.v[i] + c;
    r.v[i] = (uint32_t)t & MASK;
    c = t >> 29;
  }
  r.v[13] = (uint32_t)((int32_t)ParamsBls29::PMUL[K][13] - (int32_t)a.v[13] + c);
  // keep literal zero literal (the identity's coordinates): -0 = 0
  const bool z = a.is_zero();
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = z ? 0u : r.v[i];
  return r;
}

// out-of-line multiply / square on raw limb arrays (keeps EC routines small; see fp.hpp)
struct Fe29Raw {
  uint32_t v[14];
};
template <int TAG = 0>
NCG_MULFN Fe29Raw fe29_mul_raw(Fe29Raw a, Fe29Raw b) {
  Fe29Raw r;
  mont_mul29<ParamsBls29>(r.v, a.v, b.v);
  return r;
}
template <int TAG = 0>
NCG_MULFN Fe29Raw fe29_sqr_raw(Fe29Raw a) {
  Fe29Raw r;
  mont_sqr29<ParamsBls29>(r.v, a.v);
  return r;
}

template <int A, int B>
NCG_DI Fe29<2> operator*(const Fe29<A>& a, const Fe29<B>& b) {
  static_assert((long)A * B <= (1L << 24), "Montgomery product would exceed 2p: reduce an operand");
  Fe29Raw x, y;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    x.v[i] = a.v[i];
    y.v[i] = b.v[i];
  }
  Fe29Raw z = fe29_mul_raw(x, y);
  Fe29<2> r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = z.v[i];
  return r;
}
template <int A>
NCG_DI Fe29<2> f_sqr(const Fe29<A>& a) {
  static_assert((long)A * A <= (1L << 24), "Montgomery square would exceed 2p");
  Fe29Raw x;
#pragma unroll
  for (int i = 0; i < 14; i++) x.v[i] = a.v[i];
  Fe29Raw z = fe29_sqr_raw(x);
  Fe29<2> r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = z.v[i];
  return r;
}
template <int A>
NCG_DI Fe29<2 * A> f_dbl(const Fe29<A>& a) {
  return a + a;
}

// a*b - c*d with one Montgomery reduction (fp29.hpp mont_muladd29 on a, b, -c, d): the pattern
// Y3 = R (Q - X3) - Y1 PPP of every XYZZ / Jacobian addition.  Saves one reduction (196 of 784 multiply-adds).
template <int A, int B, int C, int D>
NCG_DI Fe29<2> f_mulsub(const Fe29<A>& a, const Fe29<B>& b, const Fe29<C>& c, const Fe29<D>& d) {
  constexpr int KC = 1 << fe29_pow2ceil_log(C);
  static_assert((long)A * B + (long)KC * D <= (1L << 24), "fused product pair would exceed 2p: reduce an operand");
  const Fe29<KC> nc = f_neg(c);
  Fe29<2> r;
  mont_muladd29<ParamsBls29>(r.v, a.v, b.v, nc.v, d.v);
  return r;
}

// canonical residue in [0, p) as 29-bit limbs (still Montgomery form): multiply by the
// Montgomery one is the identity map and lands below p + 1, then one conditional subtraction.
template <int A>
NCG_DI Fe29<1> fe29_canon(const Fe29<A>& a) {
  constexpr uint32_t MASK = (1u << 29) - 1u;
  Fe29<2> t = a * Fe29<1>::one();
  // s = t - p with borrow
  uint32_t s[14];
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    int32_t d = (int32_t)t.v[i] - (int32_t)ParamsBls29::P[i] + c;
    s[i] = (uint32_t)d & MASK;
    c = d >> 29;
  }
  Fe29<1> r;
  const bool ge = c == 0;  // no borrow: t >= p
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = ge ? s[i] : t.v[i];
  return r;
}

// a == 0 (mod p) for a value below A*p: a = j*p for some j < A.  The low limb gives j
// (j = a0 * p0^-1 mod 2^29); almost always j >= A and the test ends after one multiply.
template <int A>
NCG_DI bool f_eqz(const Fe29<A>& a) {
  constexpr uint32_t MASK = (1u << 29) - 1u;
  const uint32_t j = (0u - a.v[0] * ParamsBls29::INV) & MASK;  // INV = -p^-1
  if (j >= (uint32_t)A) return false;
  // exact: compare with j*p
  uint64_t c = 0;
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    c += (uint64_t)j * ParamsBls29::P[i];
    uint32_t limb = i < 13 ? ((uint32_t)c & MASK) : (uint32_t)c;
    diff |= limb ^ a.v[i];
    c >>= 29;
  }
  return diff == 0;
}
template <int A, int B>
NCG_DI bool f_eq(const Fe29<A>& a, const Fe29<B>& b) {
  return f_eqz(a - b);
}

// ---- powers with a fixed public exponent (inversion, square roots): sliding windows of W bits over a
// schedule computed at compile time.  (p - 3) / 4 and p - 2 have 379 / 381 bits and 228 / 229 of them set:
// square-and-multiply is 378 S + 227 M, windows of 4 bits 378 S + 78 M + the 8-entry table of odd powers
// (27 % fewer multiply-adds), of 3 bits 378 S + 105 M + 4 entries (23 %; 56 registers less).  The exponent is
// the same for every lane, so the table index is wave-uniform and the entry is picked by scalar branches.
struct PowSched {
  int n;             // steps: `nsq[k]` squarings, then a multiplication by a^(2 idx[k] + 1); step 0 only loads
  int trail;         // squarings after the last multiplication
  uint8_t nsq[160];
  uint8_t idx[160];
};
template <int W>
constexpr PowSched make_pow_sched(const uint32_t (&e)[12], uint32_t minus) {
  uint32_t w[12] = {};
  for (int i = 0; i < 12; i++) w[i] = e[i];
  w[0] -= minus;  // callers pass exponents whose low word does not borrow
  PowSched s{};
  int i = 383;
  while (i >= 0 && !((w[i >> 5] >> (i & 31)) & 1u)) i--;
  int pend = 0;
  while (i >= 0) {
    if (!((w[i >> 5] >> (i & 31)) & 1u)) {
      pend++;
      i--;
      continue;
    }
    int j = i - W + 1 < 0 ? 0 : i - W + 1;
    while (!((w[j >> 5] >> (j & 31)) & 1u)) j++;
    uint32_t v = 0;
    for (int b = i; b >= j; b--) v = (v << 1) | ((w[b >> 5] >> (b & 31)) & 1u);
    s.nsq[s.n] = (uint8_t)(pend + (i - j + 1));
    s.idx[s.n] = (uint8_t)((v - 1) >> 1);
    s.n++;
    pend = 0;
    i = j - 1;
  }
  s.trail = pend;
  return s;
}
template <int W>
struct BlsPowSched {
  static constexpr PowSched SQRT_M1 = make_pow_sched<W>(BlsFpConsts::SQRT_EXP_M1, 0u);  // (p - 3) / 4
  static constexpr PowSched SQRT = make_pow_sched<W>(BlsFpConsts::SQRT_EXP, 0u);        // (p + 1) / 4
  static constexpr PowSched INV = make_pow_sched<W>(BlsFpConsts::P32, 2u);              // p - 2
};
// W = 3: the four odd powers stay in named registers (an indexed array would live in scratch memory)
constexpr int NCG_POW_W = 3;
NCG_DI Fe29<2> fe29_pow_sched(const Fe29<2>& a, const PowSched& s) {
  const Fe29<2> a2 = f_sqr(a);
  const Fe29<2> t3 = a * a2, t5 = t3 * a2, t7 = t5 * a2;
  auto pick = [&](int idx) -> Fe29<2> {
    switch (idx) {
      case 0: return a;
      case 1: return t3;
      case 2: return t5;
      default: return t7;
    }
  };
  Fe29<2> r = pick(s.idx[0]);
  for (int k = 1; k < s.n; k++) {
    const int nsq = s.nsq[k];
    for (int q = 0; q < nsq; q++) r = f_sqr(r);
    r = r * pick(s.idx[k]);
  }
  for (int q = 0; q < s.trail; q++) r = f_sqr(r);
  return r;
}

// Fermat inversion a^(p - 2) (value of modular.ts:159-182 invert; 0 -> 0)
template <int A>
NCG_DI Fe29<2> f_inv(const Fe29<A>& a) {
  return fe29_pow_sched(Fe29<2>(a * Fe29<1>::one()), BlsPowSched<NCG_POW_W>::INV);
}

// ---- wire format (12 x 32-bit LE limbs, canonical residue) <-> Fe29 (Montgomery)
NCG_DI Fe29<2> fe29_from_wire(const uint32_t* __restrict__ p) {
  constexpr uint32_t MASK = (1u << 29) - 1u;
  uint32_t w[13];
#pragma unroll
  for (int i = 0; i < 12; i++) w[i] = p[i];
  w[12] = 0;
  Fe29<16> t;  // any 384-bit value is below 2^384 < 16 p
#pragma unroll
  for (int i = 0; i < 14; i++) {
    const int bit = 29 * i, limb = bit >> 5, sh = bit & 31;  // limb <= 11
    uint64_t two = ((uint64_t)w[limb + 1] << 32) | w[limb];
    t.v[i] = (uint32_t)(two >> sh) & MASK;
  }
  Fe29<1> r2;
#pragma unroll
  for (int i = 0; i < 14; i++) r2.v[i] = ParamsBls29::R2[i];
  return t * r2;
}
template <int A>
NCG_DI void fe29_to_wire(uint32_t* __restrict__ p, const Fe29<A>& a) {
  // out of Montgomery form: multiply by the integer 1, then canonicalise
  Fe29<1> one_int = Fe29<1>::zero();
  one_int.v[0] = 1;
  Fe29<2> t = a * one_int;  // < a/R + p <= p  (a < 2^12 p << R)
  constexpr uint32_t MASK = (1u << 29) - 1u;
  uint32_t s[14];
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    int32_t d = (int32_t)t.v[i] - (int32_t)ParamsBls29::P[i] + c;
    s[i] = (uint32_t)d & MASK;
    c = d >> 29;
  }
  const bool ge = c == 0;
  uint32_t l[15];
#pragma unroll
  for (int i = 0; i < 14; i++) l[i] = ge ? s[i] : t.v[i];
  l[14] = 0;
  // repack 14 x 29 -> 12 x 32
#pragma unroll
  for (int k = 0; k < 12; k++) {
    const int bit = 32 * k, limb = bit / 29, sh = bit % 29;
    uint64_t acc = (uint64_t)l[limb] >> sh;
    acc |= (uint64_t)l[limb + 1] << (29 - sh);
    if (limb + 2 <= 14) acc |= (uint64_t)l[limb + 2] << (58 - sh);
    p[k] = (uint32_t)acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Fp2 = Fp[u]/(u^2+1) over Fe29 (G2).  Values of `_Field2` ops, src/abstract/tower.ts:393-438.
template <int B>
struct Fe29x2 {
  Fe29<B> c0, c1;
  Fe29x2() = default;
  NCG_DI Fe29x2(const Fe29<B>& a, const Fe29<B>& b) : c0(a), c1(b) {}
  template <int B2, class = typename std::enable_if<(B2 < B)>::type>
  NCG_DI Fe29x2(const Fe29x2<B2>& o) : c0(o.c0), c1(o.c1) {}
  static NCG_DI Fe29x2 zero() { return {Fe29<B>::zero(), Fe29<B>::zero()}; }
  static NCG_DI Fe29x2 one() { return {Fe29<B>::one(), Fe29<B>::zero()}; }
  NCG_DI bool is_zero() const { return c0.is_zero() && c1.is_zero(); }  // literal
};
template <int A, int B>
NCG_DI Fe29x2<A + B> operator+(const Fe29x2<A>& a, const Fe29x2<B>& b) {  // tower.ts:404
  return {a.c0 + b.c0, a.c1 + b.c1};
}
template <int A, int B>
NCG_DI Fe29x2<A + (1 << fe29_pow2ceil_log(B))> operator-(const Fe29x2<A>& a, const Fe29x2<B>& b) {  // :413
  return {a.c0 - b.c0, a.c1 - b.c1};
}
template <int A, int B>
NCG_DI Fe29x2<6> operator*(const Fe29x2<A>& a, const Fe29x2<B>& b) {  // tower.ts:420-431 (Karatsuba)
  auto t1 = a.c0 * b.c0;
  auto t2 = a.c1 * b.c1;
  auto m = (a.c0 + a.c1) * (b.c0 + b.c1);
  Fe29<6> o0 = t1 - t2;          // bound 4
  Fe29<6> o1 = m - (t1 + t2);    // bound 6
  return {o0, o1};
}
template <int A>
NCG_DI Fe29x2<2> f_sqr(const Fe29x2<A>& a) {  // tower.ts:432-438
  auto s = a.c0 + a.c1;
  auto d = a.c0 - a.c1;
  auto c = a.c0 + a.c0;
  return {s * d, c * a.c1};
}
template <int A>
NCG_DI Fe29x2<2 * A> f_dbl(const Fe29x2<A>& a) {
  return {a.c0 + a.c0, a.c1 + a.c1};
}
template <int A>
NCG_DI Fe29x2<(1 << fe29_pow2ceil_log(A))> f_neg(const Fe29x2<A>& a) {  // tower.ts:393
  return {f_neg(a.c0), f_neg(a.c1)};
}
template <int A>
NCG_DI bool f_eqz(const Fe29x2<A>& a) {
  return f_eqz(a.c0) && f_eqz(a.c1);
}
template <int A>
NCG_DI Fe29x2<2> f_inv(const Fe29x2<A>& a) {  // tower.ts:458-475
  auto f = f_inv(f_sqr(a.c0) + f_sqr(a.c1));
  return {f * a.c0, f * f_neg(a.c1)};
}

// ---------------------------------------------------------------------------------------------
// Lane-paired Fp2: ONE element lives on two adjacent lanes (even lane: c0, odd lane: c1), so a
// G2 point costs each lane the registers of a G1 point - the unpaired form above needs ~2x the
// registers and runs the heavy kernels at 1 wave/SIMD with accumulator-register spills.  The
// partner's half arrives by DPP quad_perm [1,0,3,2] (a full-rate VALU move, no LDS).  Same mad
// count as Karatsuba: mul = 2 lanes x (2 products + 1 reduction), sqr = 2 lanes x 1 product.
// Every data-dependent decision (is_zero, f_eqz) is made pair-uniform, so the two lanes of a
// pair always follow the same control flow.  Device only.
NCG_DI uint32_t pair_swap(uint32_t v) {
#ifdef __HIP_DEVICE_COMPILE__
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
#else
  return v;
#endif
}
NCG_DI bool pair_odd() {
#ifdef __HIP_DEVICE_COMPILE__
  return (threadIdx.x & 1u) != 0;
#else
  return false;
#endif
}
template <int B>
NCG_DI Fe29<B> pair_swap(const Fe29<B>& a) {
  Fe29<B> r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = pair_swap(a.v[i]);
  return r;
}
template <int B>
NCG_DI Fe29<B> fe29_select(bool c, const Fe29<B>& a, const Fe29<B>& b) {  // c ? a : b
  Fe29<B> r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}

// Paired product, out of line: takes only this lane's halves; the partner's halves are fetched
// inside, phase by phase, so the caller keeps 28 argument registers live instead of 56.
//   even lane: a0 b0 + (2^K p - a1) b1      odd lane: a0 b1 + a1 b0
template <int K>
NCG_MULFN Fe29Raw fe29x2p_mul_raw(Fe29Raw a, Fe29Raw b) {
  Fe29Raw r;
#ifdef __HIP_DEVICE_COMPILE__
  constexpr int N = 14;
  constexpr uint32_t MASK = (1u << 29) - 1u;
  const bool odd = pair_odd();
#if NCG_FE29_COLS_PAIRED
  {
    uint32_t xs[2][N], ys[2][N];
    int32_t cy = 0;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t pa = pair_swap(a.v[i]);
      any |= pa;
      int32_t dd = (int32_t)ParamsBls29::PMUL[K][i] - (int32_t)pa + cy;
      const uint32_t neg = i < N - 1 ? ((uint32_t)dd & MASK) : (uint32_t)dd;
      cy = i < N - 1 ? (dd >> 29) : 0;
      xs[0][i] = odd ? pa : a.v[i];
      ys[0][i] = b.v[i];
      xs[1][i] = odd ? a.v[i] : neg;
      ys[1][i] = pair_swap(b.v[i]);
    }
    if (!odd && any == 0) {
#pragma unroll
      for (int i = 0; i < N; i++) xs[1][i] = 0;
    }
    mont_cols29<ParamsBls29, 2>(r.v, xs, ys);
    return r;
  }
#endif
  uint64_t t[2 * N];
#pragma unroll
  for (int k = 0; k < 2 * N; k++) t[k] = 0;
  {  // phase 1: (odd ? a_partner : a_own) * b_own
    uint32_t x[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t pa = pair_swap(a.v[i]);
      x[i] = odd ? pa : a.v[i];
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
      for (int j = 0; j < N; j++) t[i + j] += (uint64_t)x[i] * b.v[j];
    }
  }
  {  // phase 2: (odd ? a_own : -a_partner) * b_partner
    uint32_t z[N], w[N];
    int32_t c = 0;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t pa = pair_swap(a.v[i]);
      any |= pa;
      int32_t d = (int32_t)ParamsBls29::PMUL[K][i] - (int32_t)pa + c;
      const uint32_t neg = i < N - 1 ? ((uint32_t)d & MASK) : (uint32_t)d;
      c = i < N - 1 ? (d >> 29) : 0;
      z[i] = odd ? a.v[i] : neg;
      w[i] = pair_swap(b.v[i]);
    }
    if (!odd && any == 0) {  // -0 = 0 (keeps products of literal zeros literal)
#pragma unroll
      for (int i = 0; i < N; i++) z[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
      for (int j = 0; j < N; j++) t[i + j] += (uint64_t)z[i] * w[j];
    }
  }
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    uint64_t T = t[k] + carry;
    uint32_t q = ((uint32_t)T * ParamsBls29::INV) & MASK;
    T += (uint64_t)q * (uint32_t)ParamsBls29::P[0];
    carry = T >> 29;
#pragma unroll
    for (int j = 1; j < N; j++) t[k + j] += (uint64_t)q * (uint32_t)ParamsBls29::P[j];
  }
#pragma unroll
  for (int k = N; k < 2 * N; k++) {
    uint64_t T = t[k] + carry;
    r.v[k - N] = (uint32_t)T & MASK;
    carry = T >> 29;
  }
#else
  for (int i = 0; i < 14; i++) r.v[i] = 0;
#endif
  return r;
}

template <int B>
struct Fe29x2P {
  Fe29<B> h;  // this lane's component
  static constexpr int BOUND = B;
  Fe29x2P() = default;
  NCG_DI explicit Fe29x2P(const Fe29<B>& a) : h(a) {}
  template <int B2, class = typename std::enable_if<(B2 < B)>::type>
  NCG_DI Fe29x2P(const Fe29x2P<B2>& o) : h(o.h) {}
  static NCG_DI Fe29x2P zero() { return Fe29x2P(Fe29<B>::zero()); }
  static NCG_DI Fe29x2P one() { return Fe29x2P(fe29_select(pair_odd(), Fe29<B>::zero(), Fe29<B>::one())); }
  NCG_DI bool is_zero() const {  // literal, both halves
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) o |= h.v[i];
    o |= pair_swap(o);
    return o == 0;
  }
};
template <int A, int B>
NCG_DI Fe29x2P<A + B> operator+(const Fe29x2P<A>& a, const Fe29x2P<B>& b) {
  return Fe29x2P<A + B>(a.h + b.h);
}
template <int A, int B>
NCG_DI Fe29x2P<A + (1 << fe29_pow2ceil_log(B))> operator-(const Fe29x2P<A>& a, const Fe29x2P<B>& b) {
  return Fe29x2P<A + (1 << fe29_pow2ceil_log(B))>(a.h - b.h);
}
template <int A>
NCG_DI Fe29x2P<2 * A> f_dbl(const Fe29x2P<A>& a) {
  return Fe29x2P<2 * A>(a.h + a.h);
}
template <int A>
NCG_DI Fe29x2P<(1 << fe29_pow2ceil_log(A))> f_neg(const Fe29x2P<A>& a) {
  return Fe29x2P<(1 << fe29_pow2ceil_log(A))>(f_neg(a.h));
}
// (a0 + a1 u)(b0 + b1 u): even lane a0 b0 + (-a1) b1, odd lane a0 b1 + a1 b0  (tower.ts:420-431 value)
template <int A, int B>
NCG_DI Fe29x2P<2> operator*(const Fe29x2P<A>& a, const Fe29x2P<B>& b) {
  constexpr int K = fe29_pow2ceil_log(A);
  static_assert(4L * (1 << K) * B <= (1L << 25), "paired Fp2 product would exceed 2p: reduce an operand");
  Fe29Raw x, y;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    x.v[i] = a.h.v[i];
    y.v[i] = b.h.v[i];
  }
  Fe29Raw r = fe29x2p_mul_raw<K>(x, y);
  Fe29<2> o;
#pragma unroll
  for (int i = 0; i < 14; i++) o.v[i] = r.v[i];
  return Fe29x2P<2>(o);
}
// a*b - c*d in Fp2 with ONE reduction per lane (the XYZZ pattern Y3 = R (Q - X3) - Y1 PPP): four products share
// the column registers.  56 product terms + 14 reduction terms per column stay below 2^64 only because limb 13 of
// any value below 2^12 p is at most 53256: the exact worst case over all columns is 0.922 * 2^64 (limbs 0..12 at
// 2^29 - 1, limb 13 at its maximum, q at 2^29 - 1).
//   even lane: a0 b0 + (-a1) b1 + (-c0) d0 + c1 d1        odd lane: a0 b1 + a1 b0 + (-c0) d1 + (-c1) d0
template <int A, int B, int C, int D>
NCG_DI Fe29x2P<2> f_mulsub(const Fe29x2P<A>& a, const Fe29x2P<B>& b, const Fe29x2P<C>& c, const Fe29x2P<D>& d) {
  constexpr int KA = fe29_pow2ceil_log(A), KC = fe29_pow2ceil_log(C);
  static_assert(4L * (1 << KA) * B + 4L * (1 << KC) * D <= (1L << 25), "fused paired products would exceed 2p: reduce an operand");
  static_assert((1 << KA) <= 4096 && (1 << KC) <= 4096 && B <= 4096 && D <= 4096, "operand bound above 2^12 p: column bound unproven");
  Fe29<2> o;
#ifdef __HIP_DEVICE_COMPILE__
  constexpr int N = 14;
  constexpr uint32_t MASK = (1u << 29) - 1u;
  const bool odd = pair_odd();
  const Fe29<(1 << KA)> na = f_neg(a.h);  // 2^KA p - own a
  const Fe29<(1 << KC)> nc = f_neg(c.h);  // 2^KC p - own c
#if NCG_FE29_COLS_PAIRED
  {
    uint32_t xs[4][N], ys[4][N];
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t pa = pair_swap(a.h.v[i]), pna = pair_swap(na.v[i]), pnc = pair_swap(nc.v[i]), pc = pair_swap(c.h.v[i]);
      xs[0][i] = odd ? pa : a.h.v[i];
      ys[0][i] = b.h.v[i];
      xs[1][i] = odd ? a.h.v[i] : pna;
      ys[1][i] = pair_swap(b.h.v[i]);
      xs[2][i] = odd ? pnc : nc.v[i];
      ys[2][i] = d.h.v[i];
      xs[3][i] = odd ? nc.v[i] : pc;
      ys[3][i] = pair_swap(d.h.v[i]);
    }
    mont_cols29<ParamsBls29, 4>(o.v, xs, ys);
    return Fe29x2P<2>(o);
  }
#endif
  uint64_t t[2 * N];
#pragma unroll
  for (int k = 0; k < 2 * N; k++) t[k] = 0;
  {  // (odd ? a_partner : a_own) * b_own
    uint32_t x[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t pa = pair_swap(a.h.v[i]);
      x[i] = odd ? pa : a.h.v[i];
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
      for (int j = 0; j < N; j++) t[i + j] += (uint64_t)x[i] * b.h.v[j];
    }
  }
  {  // (odd ? a_own : -a_partner) * b_partner
    uint32_t z[N], w[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t pna = pair_swap(na.v[i]);
      z[i] = odd ? a.h.v[i] : pna;
      w[i] = pair_swap(b.h.v[i]);
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
      for (int j = 0; j < N; j++) t[i + j] += (uint64_t)z[i] * w[j];
    }
  }
  {  // (odd ? -c_partner : -c_own) * d_own
    uint32_t x[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t pnc = pair_swap(nc.v[i]);
      x[i] = odd ? pnc : nc.v[i];
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
      for (int j = 0; j < N; j++) t[i + j] += (uint64_t)x[i] * d.h.v[j];
    }
  }
  {  // (odd ? -c_own : c_partner) * d_partner
    uint32_t z[N], w[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t pc = pair_swap(c.h.v[i]);
      z[i] = odd ? nc.v[i] : pc;
      w[i] = pair_swap(d.h.v[i]);
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
      for (int j = 0; j < N; j++) t[i + j] += (uint64_t)z[i] * w[j];
    }
  }
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    uint64_t T = t[k] + carry;
    uint32_t q = ((uint32_t)T * ParamsBls29::INV) & MASK;
    T += (uint64_t)q * (uint32_t)ParamsBls29::P[0];
    carry = T >> 29;
#pragma unroll
    for (int j = 1; j < N; j++) t[k + j] += (uint64_t)q * (uint32_t)ParamsBls29::P[j];
  }
#pragma unroll
  for (int k = N; k < 2 * N; k++) {
    uint64_t T = t[k] + carry;
    o.v[k - N] = (uint32_t)T & MASK;
    carry = T >> 29;
  }
#else
  for (int i = 0; i < 14; i++) o.v[i] = 0;
#endif
  return Fe29x2P<2>(o);
}

// even lane (a0 + a1)(a0 - a1), odd lane (a0 + a0) a1  (tower.ts:432-438 value)
template <int A>
NCG_DI Fe29x2P<2> f_sqr(const Fe29x2P<A>& a) {
  constexpr int KA = 1 << fe29_pow2ceil_log(A);
  static_assert(2L * A * (A + KA) <= (1L << 24), "paired Fp2 square would exceed 2p");
  const bool odd = pair_odd();
  const Fe29<A> pa = pair_swap(a.h);
  const Fe29<A> a0 = fe29_select(odd, pa, a.h), a1 = fe29_select(odd, a.h, pa);
  const Fe29<2 * A> s = a0 + fe29_select(odd, a0, a1);
  const Fe29<A + KA> d = a0 - a1;
  const Fe29<A + KA> y = fe29_select(odd, Fe29<A + KA>(a1), d);
#if NCG_FE29_COLS_PAIRED && defined(__HIP_DEVICE_COMPILE__)
  {  // the lane's one product in the column-wise form too (same value as s * y)
    static_assert(2L * A * (A + KA) <= (1L << 24), "operand bounds");
    uint32_t xs[1][14], ys[1][14];
#pragma unroll
    for (int i = 0; i < 14; i++) {
      xs[0][i] = s.v[i];
      ys[0][i] = y.v[i];
    }
    Fe29<2> o;
    mont_cols29<ParamsBls29, 1>(o.v, xs, ys);
    return Fe29x2P<2>(o);
  }
#endif
  return Fe29x2P<2>(s * y);
}
template <int A>
NCG_DI bool f_eqz(const Fe29x2P<A>& a) {
  uint32_t e = f_eqz(a.h) ? 1u : 0u;
  e &= pair_swap(e);
  return e != 0;
}
template <int A>
NCG_DI Fe29x2P<2> f_inv(const Fe29x2P<A>& a) {  // conj(a) / (a0^2 + a1^2)  (tower.ts:458-475 value)
  constexpr int KA = 1 << fe29_pow2ceil_log(A);
  Fe29<2> n = f_sqr(a.h);
  Fe29<2> inv = f_inv(n + pair_swap(n));
  const Fe29<KA> num = fe29_select(pair_odd(), f_neg(a.h), Fe29<KA>(a.h));
  return Fe29x2P<2>(inv * num);
}

// storage types used by the curve templates (every stored coordinate is below 64 p)
using FeBls = Fe29<64>;
using FeBls2 = Fe29x2<64>;
using FeBls2P = Fe29x2P<64>;

}  // namespace ncg
