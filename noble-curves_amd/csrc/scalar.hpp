// Scalar-side helpers: small multi-limb integer ops, secp256k1 GLV decomposition and the
// signed-odd fixed-window recoding used by the batch ladders.
//
// GLV: the reference's `_splitEndoScalar` (src/abstract/weierstrass.ts:121-148) computes
//   c1 = round(b2*k/n), c2 = round(-b1*k/n), k1 = k - c1*a1 - c2*a2, k2 = -c1*b1 - c2*b2
// with the basis of src/secp256k1.ts:58-64.  Here the two divisions by n are replaced by
// multiplications with 384-bit reciprocals g1 = round(2^384*b2/n), g2 = round(2^384*(-b1)/n);
// c1/c2 may differ from the exact roundings by one unit, which still yields a lattice-exact
// decomposition k = k1 + lambda*k2 (mod n) with |k1|,|k2| < 2^128 - and since secp256k1 has
// cofactor 1 the resulting group element is identical (SURVEY 8a gotcha 1).
#pragma once
#include "fp.hpp"

namespace ncg {

// r[0..NA+NB) = a * b (schoolbook, little-endian 32-bit limbs)
template <int NA, int NB>
NCG_DI void mp_mul(uint32_t (&r)[NA + NB], const uint32_t (&a)[NA], const uint32_t (&b)[NB]) {
#pragma unroll
  for (int i = 0; i < NA + NB; i++) r[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + c;
      r[i + j] = (uint32_t)t;
      c = (uint32_t)(t >> 32);
    }
    r[i + NB] = c;
  }
}

template <int N>
NCG_DI uint32_t mp_add(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = __builtin_addc(a[i], b[i], c, &c);
  return c;
}
template <int N>
NCG_DI uint32_t mp_sub(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = __builtin_subc(a[i], b[i], c, &c);
  return c;
}
template <int N>
NCG_DI void mp_neg(uint32_t (&a)[N]) {  // two's complement negate
  uint32_t c = 1;
#pragma unroll
  for (int i = 0; i < N; i++) a[i] = __builtin_addc(~a[i], 0u, c, &c);
}
template <int N>
NCG_DI bool mp_is_zero(const uint32_t (&a)[N]) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < N; i++) o |= a[i];
  return o == 0;
}
// a <<= s (0 < s < 32)
template <int N>
NCG_DI void mp_shl(uint32_t (&a)[N], int s) {
#pragma unroll
  for (int i = N - 1; i > 0; i--) a[i] = (a[i] << s) | (a[i - 1] >> (32 - s));
  a[0] <<= s;
}

// |k1|, |k2| as 5-limb magnitudes plus sign flags.
struct GlvSplit {
  uint32_t k1[5], k2[5];
  bool k1neg, k2neg;
};

// round(k * g / 2^384) for 256-bit k and g: limbs 12..15 of the product plus the rounding bit.
NCG_DI void glv_mul_shift_384(uint32_t (&c)[4], const uint32_t (&k)[8], const uint32_t (&g)[8]) {
  uint32_t prod[16];
  mp_mul<8, 8>(prod, k, g);
  uint32_t carry = prod[11] >> 31;
  uint32_t cy = 0;
  c[0] = __builtin_addc(prod[12], carry, 0u, &cy);
#pragma unroll
  for (int i = 1; i < 4; i++) c[i] = __builtin_addc(prod[12 + i], 0u, cy, &cy);
}

NCG_DI GlvSplit secp_glv_split(const uint32_t (&k)[8]) {
  uint32_t g1[8], g2[8], a1[5], mb1[5], a2[5], b2[5];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    g1[i] = SecpGlv::G1[i];
    g2[i] = SecpGlv::G2[i];
  }
#pragma unroll
  for (int i = 0; i < 5; i++) {
    a1[i] = SecpGlv::A1[i];
    mb1[i] = SecpGlv::MB1[i];
    a2[i] = SecpGlv::A2[i];
    b2[i] = SecpGlv::B2[i];
  }
  uint32_t c1[4], c2[4];
  glv_mul_shift_384(c1, k, g1);
  glv_mul_shift_384(c2, k, g2);
  // k1 = k - c1*a1 - c2*a2 ; k2 = c1*(-b1) - c2*b2   (9-limb two's complement)
  uint32_t p1[9], p2[9], t[9], kk[9];
  mp_mul<4, 5>(p1, c1, a1);
  mp_mul<4, 5>(p2, c2, a2);
  mp_add<9>(t, p1, p2);
#pragma unroll
  for (int i = 0; i < 8; i++) kk[i] = k[i];
  kk[8] = 0;
  uint32_t r1[9], r2[9];
  mp_sub<9>(r1, kk, t);
  mp_mul<4, 5>(p1, c1, mb1);
  mp_mul<4, 5>(p2, c2, b2);
  mp_sub<9>(r2, p1, p2);
  GlvSplit s;
  s.k1neg = (r1[8] >> 31) != 0;
  s.k2neg = (r2[8] >> 31) != 0;
  if (s.k1neg) mp_neg<9>(r1);
  if (s.k2neg) mp_neg<9>(r2);
#pragma unroll
  for (int i = 0; i < 5; i++) {
    s.k1[i] = r1[i];
    s.k2[i] = r2[i];
  }
  return s;
}

// Signed-odd fixed-window recoding (Joye-Tunstall style, closed form).
// For an odd integer k < 2^L (L = M*W) write k = sum_{i<M} d_i 2^(W i) with every d_i odd,
// |d_i| <= 2^W - 1:  d_i = 2*((k~ >> (W i + 1)) & (2^W - 1)) - (2^W - 1),  k~ = k | 2^L.
// `SignedOddWindows` holds (k~ >> 1) top-aligned in NL limbs and pops windows MSB first, so
// all limb indexing is static (no scratch).  An even k is made odd by adding 1; the caller
// subtracts the point once at the end (`was_even`).
template <int NL, int W, int M>
struct SignedOddWindows {
  static_assert(W * M <= 32 * NL, "window register too small");
  uint32_t r[NL];
  bool was_even;
  // k given as NK <= NL limbs, must satisfy k + 1 < 2^(W*M)
  template <int NK>
  NCG_DI void init(const uint32_t (&k)[NK]) {
    uint32_t t[NL + 1];
#pragma unroll
    for (int i = 0; i < NL + 1; i++) t[i] = i < NK ? k[i] : 0u;
    was_even = (t[0] & 1u) == 0;
    if (was_even) {  // k + 1 (k even: no carry out of bit 0)
      t[0] |= 1u;
    }
    // set bit L, shift right by 1, then top-align the L-bit value in NL limbs
    constexpr int L = W * M;
    t[L / 32] |= 1u << (L % 32);
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = (t[i] >> 1) | (t[i + 1] << 31);
    constexpr int SH = 32 * NL - L;  // left shift to top-align
    constexpr int LS = SH / 32, BS = SH % 32;
    if (LS > 0) {
#pragma unroll
      for (int i = NL - 1; i >= 0; i--) r[i] = (i - LS >= 0) ? r[i - LS] : 0u;
    }
    if (BS > 0) mp_shl<NL>(r, BS);
  }
  // next window (MSB first): returns signed odd digit
  NCG_DI int pop() {
    uint32_t b = r[NL - 1] >> (32 - W);
    mp_shl<NL>(r, W);
    return 2 * (int)b - ((1 << W) - 1);
  }
};

}  // namespace ncg
