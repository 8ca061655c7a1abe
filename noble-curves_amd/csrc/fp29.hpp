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

// r = a*b*R^-1 mod p (R = 2^(29N)); inputs: limbs < 2^29, values < 2^12 p; output limbs < 2^29,
// value < a*b/R + p.
template <class PR>
NCG_DI void mont_mul29(uint32_t (&r)[PR::N], const uint32_t (&a)[PR::N], const uint32_t (&b)[PR::N]) {
#ifndef __HIP_DEVICE_COMPILE__
  mont_mul58_host<PR>(r, a, b);
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
