// Half-size scalars for the ed25519 verification equation (Antipa, Brown, Gallant, Lambert, Struik, Vanstone,
// "Accelerated verification of ECDSA signatures", SAC 2005, carried over to the cofactored EdDSA check).
//
// The reference accepts iff [8] X == O with X = [s]B - [k]A - R (src/abstract/edwards.ts:985-988).  For any integer
// u with 0 < |u| < L,  [8] X == O  <=>  [u][8] X == O  ([8]X lies in the subgroup of prime order L), and with
// v = u k (mod L):   [u][8] X = [8]([u s mod L] B - [v] A - [u] R)   - the multiples of L that the reductions drop
// act on A and B through their 8-torsion parts only, which the factor 8 removes.  The extended Euclidean
// algorithm on (L, k), stopped at the first remainder below 2^127, gives such a pair with 0 <= v < 2^127 and
// |u| < 2^126 (|t_i| r_{i-1} <= L): the doubling chain shrinks from 253 to 128 steps, and the 253-bit fixed-base
// scalar u s mod L is split over two precomputed tables (B and 2^128 B).
//
// ed_halve_scalar runs the Euclidean steps as shift-and-subtract (r0 -= r1 << s with the largest s that fits, so r0
// loses at least one bit per iteration): ~108 iterations on average, 180 for the golden-ratio worst case, <= 253
// always; per-lane trip counts differ, the wavefront runs to the slowest lane.
#pragma once
#include "scalar.hpp"
#include "sha512.hpp"

namespace ncg {

struct EdHalf {
  uint32_t u[4], v[4];  // (uneg ? -u : u) * k == v (mod L), 0 < u < 2^126, 0 <= v < 2^127
  bool uneg;
};

NCG_DI int mp_bitlen8(const uint32_t (&a)[8]) {
  int bl = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int c = __builtin_clz(a[i] | 1u);
    bl = a[i] ? 32 * i + 32 - c : bl;
  }
  return bl;
}

// k: any 256-bit value (reduced mod L first; the reference's k is already below L, edwards.ts:900-906)
NCG_DI EdHalf ed_halve_scalar(const uint32_t (&k_in)[8]) {
  uint32_t r0[8], r1[8], t0[4] = {0u, 0u, 0u, 0u}, t1[4] = {1u, 0u, 0u, 0u};
  {
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = i < 8 ? k_in[i] : 0u;
    mod_l_512(r1, x);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) r0[i] = (uint32_t)Orders::ED_L[i];
  bool odd = false;
  while ((r1[4] | r1[5] | r1[6] | r1[7] | (r1[3] >> 31)) != 0u) {  // r1 >= 2^127; invariant r0 > r1
    const int s = mp_bitlen8(r0) - mp_bitlen8(r1);  // >= 0
    // x = r1 << s, y = t1 << s; if x > r0 use x / 2 and y / 2 (then s >= 1): r0 - x/2 = (r0 - x) + x/2
    const int bs = s & 31;
    uint32_t x[8], y[4], d[8];
#pragma unroll
    for (int i = 7; i >= 0; i--) {
      const uint64_t two = ((uint64_t)r1[i] << 32) | (i > 0 ? r1[i - 1] : 0u);
      x[i] = (uint32_t)((two << bs) >> 32);
    }
#pragma unroll
    for (int i = 3; i >= 0; i--) {
      const uint64_t two = ((uint64_t)t1[i] << 32) | (i > 0 ? t1[i - 1] : 0u);
      y[i] = (uint32_t)((two << bs) >> 32);
    }
    for (int j = s >> 5; j > 0; j--) {  // whole limbs: only when r1 is more than 32 bits shorter (rare)
#pragma unroll
      for (int i = 7; i > 0; i--) x[i] = x[i - 1];
      x[0] = 0u;
#pragma unroll
      for (int i = 3; i > 0; i--) y[i] = y[i - 1];
      y[0] = 0u;
    }
    const bool over = mp_sub<8>(d, r0, x) != 0;
    const uint32_t m = over ? 0xffffffffu : 0u;
    uint32_t h[8], g[4];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = ((x[i] >> 1) | (i < 7 ? (x[i + 1] << 31) : 0u)) & m;
    mp_add<8>(r0, d, h);
#pragma unroll
    for (int i = 0; i < 4; i++) g[i] = over ? ((y[i] >> 1) | (i < 3 ? (y[i + 1] << 31) : 0u)) : y[i];
    mp_add<4>(t0, t0, g);
    uint32_t dd[8];
    const bool lt = mp_sub<8>(dd, r0, r1) != 0;  // r0 < r1: the step is complete, exchange the rows
    const uint32_t sw = lt ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t z = (r0[i] ^ r1[i]) & sw;
      r0[i] ^= z;
      r1[i] ^= z;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t z = (t0[i] ^ t1[i]) & sw;
      t0[i] ^= z;
      t1[i] ^= z;
    }
    odd = odd != lt;
  }
  EdHalf out;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    out.u[i] = t1[i];
    out.v[i] = r1[i];
  }
  out.uneg = odd;
  return out;
}

// w = u * s mod L (u < 2^128, s < 2^256)
NCG_DI void ed_mul_mod_l(uint32_t (&w)[8], const uint32_t (&u)[4], const uint32_t (&s)[8]) {
  uint32_t p[12], x[16];
  mp_mul<4, 8>(p, u, s);
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = i < 12 ? p[i] : 0u;
  mod_l_512(w, x);
}

}  // namespace ncg
