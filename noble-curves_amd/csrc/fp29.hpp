// Radix-2^29 Montgomery multiplication with 64-bit column accumulators ("lazy carries").
//
// v_mad_u64_u32 accumulates a 32x32 product into a 64-bit register at full rate on gfx950
// (profiles/r01_ubench_instr_rates.json), but has no carry-in, so with 32-bit limbs every
// partial product costs two extra carry instructions.  With 29-bit limbs a column of up to 28
// products (14 a*b + 14 q*p) stays below 2^64, so the whole multiply is mads plus one
// shift/mask per column.
#pragma once
#include "fp.hpp"

namespace ncg {

#ifndef __HIP_DEVICE_COMPILE__
// Host twin (MSM finish, table builders): the same Montgomery product in radix 2^58 - limb pairs
// of the radix-2^29 value, same R = 2^(29N) - with 128-bit accumulators: a quarter of the
// multiplies of the 29-bit form on a 64-bit CPU.
template <class PR>
inline void mont_mul58_host(uint32_t (&r)[PR::N], const uint32_t (&a)[PR::N], const uint32_t (&b)[PR::N]) {
  constexpr int H = PR::N / 2;
  constexpr uint64_t MASK = (1ull << 58) - 1ull;
  uint64_t x[H], y[H];
  for (int i = 0; i < H; i++) {
    x[i] = (uint64_t)a[2 * i] | ((uint64_t)a[2 * i + 1] << 29);
    y[i] = (uint64_t)b[2 * i] | ((uint64_t)b[2 * i + 1] << 29);
  }
  unsigned __int128 t[2 * H];
  for (int k = 0; k < 2 * H; k++) t[k] = 0;
  for (int i = 0; i < H; i++)
    for (int j = 0; j < H; j++) t[i + j] += (unsigned __int128)x[i] * y[j];
  unsigned __int128 carry = 0;
  for (int k = 0; k < H; k++) {
    unsigned __int128 T = t[k] + carry;
    uint64_t q = ((uint64_t)T * PR::INV58) & MASK;
    T += (unsigned __int128)q * PR::P58[0];
    carry = T >> 58;
    for (int j = 1; j < H; j++) t[k + j] += (unsigned __int128)q * PR::P58[j];
  }
  for (int k = H; k < 2 * H; k++) {
    unsigned __int128 T = t[k] + carry;
    uint64_t limb = (uint64_t)T & MASK;
    r[2 * (k - H)] = (uint32_t)(limb & ((1u << 29) - 1u));
    r[2 * (k - H) + 1] = (uint32_t)(limb >> 29);
    carry = T >> 58;
  }
}
// (a*b + c*d) R^-1 with one reduction, radix 2^58 (7 x 7 x 2 + 7 x 7 terms of < 2^116 fit 128 bits)
template <class PR>
inline void mont_muladd58_host(uint32_t (&r)[PR::N], const uint32_t (&a)[PR::N], const uint32_t (&b)[PR::N],
                               const uint32_t (&c)[PR::N], const uint32_t (&d)[PR::N]) {
  constexpr int H = PR::N / 2;
  constexpr uint64_t MASK = (1ull << 58) - 1ull;
  uint64_t x[H], y[H], u[H], v[H];
  for (int i = 0; i < H; i++) {
    x[i] = (uint64_t)a[2 * i] | ((uint64_t)a[2 * i + 1] << 29);
    y[i] = (uint64_t)b[2 * i] | ((uint64_t)b[2 * i + 1] << 29);
    u[i] = (uint64_t)c[2 * i] | ((uint64_t)c[2 * i + 1] << 29);
    v[i] = (uint64_t)d[2 * i] | ((uint64_t)d[2 * i + 1] << 29);
  }
  unsigned __int128 t[2 * H];
  for (int k = 0; k < 2 * H; k++) t[k] = 0;
  for (int i = 0; i < H; i++)
    for (int j = 0; j < H; j++) t[i + j] += (unsigned __int128)x[i] * y[j] + (unsigned __int128)u[i] * v[j];
  unsigned __int128 carry = 0;
  for (int k = 0; k < H; k++) {
    unsigned __int128 T = t[k] + carry;
    uint64_t q = ((uint64_t)T * PR::INV58) & MASK;
    T += (unsigned __int128)q * PR::P58[0];
    carry = T >> 58;
    for (int j = 1; j < H; j++) t[k + j] += (unsigned __int128)q * PR::P58[j];
  }
  for (int k = H; k < 2 * H; k++) {
    unsigned __int128 T = t[k] + carry;
    uint64_t limb = (uint64_t)T & MASK;
    r[2 * (k - H)] = (uint32_t)(limb & ((1u << 29) - 1u));
    r[2 * (k - H) + 1] = (uint32_t)(limb >> 29);
    carry = T >> 58;
  }
}
#endif

// Column-wise (product-scanning) form of the same Montgomery product, for sums of NP products sharing ONE reduction:
// column k is accumulated in NP + 1 independent 64-bit chains (one per product, one for the reduction terms q_i p_(k-i)
// plus the carry) that are joined once per column - 2 (NP + 1) live accumulator registers instead of the 56 of the row-wise
// form above, whose 28 column registers stay live through the whole product.  Same multiply-add count, the same column
// bounds (every column sum is the sum the row-wise form holds in t[k]), bit-identical results.  It pays where it moves a
// kernel across a register cliff - the lane-paired G2 MSM kernels (306 -> 232 registers: two waves per SIMD instead of
// one; G2 2^18 MSM 3.71 -> 3.47 ms, verified sets 3.01 -> 2.70 ms, same box) - and costs 1-4 % elsewhere (the joins of the
// chains; G1 accumulate 211 registers without spills is 1 % SLOWER than 256 + 6 spills), so only the paired products of
// msm.o use it (NCG_FE29_COLS_PAIRED; profiles/r04_ab_cols.txt).
#ifndef NCG_FE29_COLS
#define NCG_FE29_COLS 0          // every bls12-381 base-field product of the translation unit
#endif
#ifndef NCG_FE29_COLS_PAIRED
#define NCG_FE29_COLS_PAIRED NCG_FE29_COLS   // only the lane-paired Fp2 products (fe29.hpp)
#endif
template <class PR, int NP>
NCG_DI void mont_cols29(uint32_t (&r)[PR::N], const uint32_t (&x)[NP][PR::N], const uint32_t (&y)[NP][PR::N]) {
  constexpr int N = PR::N;
  constexpr uint32_t MASK = (1u << 29) - 1u;
  uint32_t q[N];
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 2 * N; k++) {
    const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
    uint64_t s[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) s[p] = 0;
    uint64_t sq = carry;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
#pragma unroll
      for (int p = 0; p < NP; p++) s[p] += (uint64_t)x[p][i] * y[p][k - i];
      if (k >= N || i < k) sq += (uint64_t)q[i] * (uint32_t)PR::P[k - i];
    }
    uint64_t T = sq;
#pragma unroll
    for (int p = 0; p < NP; p++) T += s[p];
    if (k < N) {
      q[k] = ((uint32_t)T * PR::INV) & MASK;
      T += (uint64_t)q[k] * (uint32_t)PR::P[0];
    } else {
      r[k - N] = (uint32_t)T & MASK;
    }
    carry = T >> 29;
  }
}
// the square in the same form: off-diagonal products once against the doubled operand
template <class PR>
NCG_DI void mont_cols29_sqr(uint32_t (&r)[PR::N], const uint32_t (&a)[PR::N]) {
  constexpr int N = PR::N;
  constexpr uint32_t MASK = (1u << 29) - 1u;
  uint32_t a2[N], q[N];
#pragma unroll
  for (int i = 0; i < N; i++) a2[i] = a[i] << 1;
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 2 * N; k++) {
    const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
    uint64_t s0 = 0, s1 = 0, sq = carry;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      if (2 * i < k) {  // i < j = k - i: doubled; two chains by parity of i
        if (i & 1) s1 += (uint64_t)a2[i] * a[k - i];
        else s0 += (uint64_t)a2[i] * a[k - i];
      } else if (2 * i == k) {
        s1 += (uint64_t)a[i] * a[i];
      }
      if (k >= N || i < k) sq += (uint64_t)q[i] * (uint32_t)PR::P[k - i];
    }
    uint64_t T = sq + s0 + s1;
    if (k < N) {
      q[k] = ((uint32_t)T * PR::INV) & MASK;
      T += (uint64_t)q[k] * (uint32_t)PR::P[0];
    } else {
      r[k - N] = (uint32_t)T & MASK;
    }
    carry = T >> 29;
  }
}

// r = a*b*R^-1 mod p (R = 2^(29N)); inputs: limbs < 2^29, values < 2^12 p; output limbs < 2^29,
// value < a*b/R + p.
template <class PR>
NCG_DI void mont_mul29(uint32_t (&r)[PR::N], const uint32_t (&a)[PR::N], const uint32_t (&b)[PR::N]) {
#ifndef __HIP_DEVICE_COMPILE__
  mont_mul58_host<PR>(r, a, b);
  return;
#elif NCG_FE29_COLS
  uint32_t x[1][PR::N], y[1][PR::N];
#pragma unroll
  for (int i = 0; i < PR::N; i++) {
    x[0][i] = a[i];
    y[0][i] = b[i];
  }
  mont_cols29<PR, 1>(r, x, y);
  return;
#else
  constexpr int N = PR::N;
  constexpr uint32_t MASK = (1u << 29) - 1u;
  uint64_t t[2 * N];
#pragma unroll
  for (int k = 0; k < 2 * N; k++) t[k] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int j = 0; j < N; j++) t[i + j] += (uint64_t)a[i] * b[j];
  }
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    uint64_t T = t[k] + carry;
    uint32_t q = ((uint32_t)T * PR::INV) & MASK;
    T += (uint64_t)q * (uint32_t)PR::P[0];
    carry = T >> 29;
#pragma unroll
    for (int j = 1; j < N; j++) t[k + j] += (uint64_t)q * (uint32_t)PR::P[j];
  }
#pragma unroll
  for (int k = N; k < 2 * N; k++) {
    uint64_t T = t[k] + carry;
    r[k - N] = (uint32_t)T & MASK;
    carry = T >> 29;
  }
#endif
}

// r = (a*b + c*d)*R^-1 mod p with ONE reduction: the two products share the 64-bit column accumulators
// (28 product terms + 14 reduction terms of < 2^58 each stay below 2^64).  Output value < (a*b + c*d)/R + p.
template <class PR>
NCG_DI void mont_muladd29(uint32_t (&r)[PR::N], const uint32_t (&a)[PR::N], const uint32_t (&b)[PR::N],
                          const uint32_t (&c)[PR::N], const uint32_t (&d)[PR::N]) {
#ifndef __HIP_DEVICE_COMPILE__
  mont_muladd58_host<PR>(r, a, b, c, d);
  return;
#elif NCG_FE29_COLS
  uint32_t x[2][PR::N], y[2][PR::N];
#pragma unroll
  for (int i = 0; i < PR::N; i++) {
    x[0][i] = a[i];
    y[0][i] = b[i];
    x[1][i] = c[i];
    y[1][i] = d[i];
  }
  mont_cols29<PR, 2>(r, x, y);
  return;
#else
  constexpr int N = PR::N;
  constexpr uint32_t MASK = (1u << 29) - 1u;
  uint64_t t[2 * N];
#pragma unroll
  for (int k = 0; k < 2 * N; k++) t[k] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int j = 0; j < N; j++) t[i + j] += (uint64_t)a[i] * b[j];
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int j = 0; j < N; j++) t[i + j] += (uint64_t)c[i] * d[j];
  }
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    uint64_t T = t[k] + carry;
    uint32_t q = ((uint32_t)T * PR::INV) & MASK;
    T += (uint64_t)q * (uint32_t)PR::P[0];
    carry = T >> 29;
#pragma unroll
    for (int j = 1; j < N; j++) t[k + j] += (uint64_t)q * (uint32_t)PR::P[j];
  }
#pragma unroll
  for (int k = N; k < 2 * N; k++) {
    uint64_t T = t[k] + carry;
    r[k - N] = (uint32_t)T & MASK;
    carry = T >> 29;
  }
#endif
}

// squaring: off-diagonal products once, doubled
template <class PR>
NCG_DI void mont_sqr29(uint32_t (&r)[PR::N], const uint32_t (&a)[PR::N]) {
#ifndef __HIP_DEVICE_COMPILE__
  mont_mul58_host<PR>(r, a, a);
  return;
#elif NCG_FE29_COLS
  mont_cols29_sqr<PR>(r, a);
  return;
#else
  constexpr int N = PR::N;
  constexpr uint32_t MASK = (1u << 29) - 1u;
  uint64_t t[2 * N];
#pragma unroll
  for (int k = 0; k < 2 * N; k++) t[k] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int j = i + 1; j < N; j++) t[i + j] += (uint64_t)a[i] * a[j];
  }
#pragma unroll
  for (int k = 0; k < 2 * N; k++) t[k] <<= 1;
#pragma unroll
  for (int i = 0; i < N; i++) t[2 * i] += (uint64_t)a[i] * a[i];
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    uint64_t T = t[k] + carry;
    uint32_t q = ((uint32_t)T * PR::INV) & MASK;
    T += (uint64_t)q * (uint32_t)PR::P[0];
    carry = T >> 29;
#pragma unroll
    for (int j = 1; j < N; j++) t[k + j] += (uint64_t)q * (uint32_t)PR::P[j];
  }
#pragma unroll
  for (int k = N; k < 2 * N; k++) {
    uint64_t T = t[k] + carry;
    r[k - N] = (uint32_t)T & MASK;
    carry = T >> 29;
  }
#endif
}

}  // namespace ncg
