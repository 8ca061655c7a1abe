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

// Fermat inversion (value of modular.ts:159-182 invert); result bound 2
template <int A>
NCG_DI Fe29<2> f_inv(const Fe29<A>& a) {
  Fe29<2> base = a * Fe29<1>::one();
  Fe29<2> r = Fe29<1>::one();
  bool started = false;
  for (int w = ParamsBlsP::N - 1; w >= 0; w--) {
    uint32_t word = ParamsBlsP::P[w];
    if (w == 0) word -= 2u;
    for (int bit = 31; bit >= 0; bit--) {
      if (started) r = f_sqr(r);
      if ((word >> bit) & 1u) {
        r = started ? r * base : base;
        started = true;
      }
    }
  }
  return r;
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

// storage types used by the curve templates (every stored coordinate is below 64 p)
using FeBls = Fe29<64>;
using FeBls2 = Fe29x2<64>;

}  // namespace ncg
