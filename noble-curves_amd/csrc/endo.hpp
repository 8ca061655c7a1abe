// Scalar decomposition for MSMs over point sets KNOWN to lie in the prime-order subgroup of bls12-381
// (resident sets built by the subgroup-checking decoder, or verified once at upload).  The reference's
// pippenger (src/abstract/curve.ts:863-905) takes arbitrary curve points and has no endomorphism, so the
// default MSM path never uses this; on a verified set the group element is the same and the bucket work
// shrinks: G1 halves the window count (2 x 128-bit sub-scalars), G2 quarters it (4 x 64-bit).
//
// z = |x| = 0xd201000000010000 (BLS parameter, src/bls12-381.ts:101), r = z^4 - z^2 + 1.
//   G1: phi(x, y) = (beta x, y) acts as -z^2 on the subgroup - the identity the reference's own subgroup test
//       checks (bls12-381.ts:567-577) - so z^2 P = (beta x, -y) and k P = k1 P + k2 (z^2 P), k = k1 + k2 z^2.
//   G2: psi acts as x = -z (bls12-381.ts:599-601), so z P = -psi(P), z^2 P = psi^2(P), z^3 P = -psi^3(P) and
//       k P = sum_e d_e (z^e P), k = sum_e d_e z^e.
// Sub-scalars are balanced (|k_e| <= z^2/2 + 1, |d_e| <= z/2 + 1) with z^4 = z^2 - 1 (mod r), and come back
// as 192-bit two's-complement values; the digit kernel adds the window offset H' and needs no sign handling.
#pragma once
#include "scalar.hpp"

namespace ncg {

struct BlsEndo {
  static constexpr uint32_t Z[2] = {0x00010000u, 0xd2010000u};
  static constexpr uint32_t X2[4] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u};  // z^2, 128 bits
  static constexpr uint32_t MU_X2[5] = {0xf6cfee2eu, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x00000001u};  // floor(2^256 / z^2)
  static constexpr uint32_t MU_Z[3] = {0x56cd56b5u, 0x381204cau, 0x00000001u};                            // floor(2^128 / z)
};

// q = floor(k / z^2), t = k mod z^2 for k < 2^256 (Barrett, at most two corrections)
NCG_DI void bls_divmod_z2(uint32_t (&q)[5], uint32_t (&t)[4], const uint32_t (&k)[8]) {
  uint32_t kh[5], mu[5], xx[4];
#pragma unroll
  for (int i = 0; i < 5; i++) mu[i] = BlsEndo::MU_X2[i];
#pragma unroll
  for (int i = 0; i < 4; i++) xx[i] = BlsEndo::X2[i];
#pragma unroll
  for (int i = 0; i < 5; i++) kh[i] = (k[3 + i] >> 31) | (i < 4 ? (k[4 + i] << 1) : 0u);  // k >> 127
  uint32_t prod[10];
  mp_mul<5, 5>(prod, kh, mu);
#pragma unroll
  for (int i = 0; i < 5; i++) q[i] = (prod[4 + i] >> 1) | (prod[5 + i] << 31);  // >> 129
  uint32_t qx[9];
  mp_mul<5, 4>(qx, q, xx);
  uint32_t r5[5], k5[5], q5x[5], x5[5];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    k5[i] = k[i];
    q5x[i] = qx[i];
    x5[i] = i < 4 ? xx[i] : 0u;
  }
  mp_sub<5>(r5, k5, q5x);  // k - q z^2 < 3 z^2: the low 160 bits hold it
#pragma unroll
  for (int round = 0; round < 3; round++) {
    uint32_t d[5];
    const bool ge = mp_sub<5>(d, r5, x5) == 0;
    uint32_t cy = ge ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      r5[i] = ge ? d[i] : r5[i];
      const uint32_t s = q[i] + cy;
      cy = s < cy ? 1u : 0u;
      q[i] = s;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) t[i] = r5[i];
}

// q = floor(v / z), t = v mod z for v < 2^128
NCG_DI void bls_divmod_z(uint32_t (&q)[3], uint32_t (&t)[2], const uint32_t (&v)[4]) {
  uint32_t vh[3], mu[3], zz[2];
#pragma unroll
  for (int i = 0; i < 3; i++) mu[i] = BlsEndo::MU_Z[i];
  zz[0] = BlsEndo::Z[0];
  zz[1] = BlsEndo::Z[1];
  vh[0] = (v[1] >> 31) | (v[2] << 1);  // v >> 63
  vh[1] = (v[2] >> 31) | (v[3] << 1);
  vh[2] = v[3] >> 31;
  uint32_t prod[6];
  mp_mul<3, 3>(prod, vh, mu);
  q[0] = (prod[2] >> 1) | (prod[3] << 31);  // >> 65
  q[1] = (prod[3] >> 1) | (prod[4] << 31);
  q[2] = (prod[4] >> 1) | (prod[5] << 31);
  uint32_t qz[5];
  mp_mul<3, 2>(qz, q, zz);
  uint32_t r3[3], v3[3] = {v[0], v[1], v[2]}, qz3[3] = {qz[0], qz[1], qz[2]}, z3[3] = {zz[0], zz[1], 0u};
  mp_sub<3>(r3, v3, qz3);
#pragma unroll
  for (int round = 0; round < 3; round++) {
    uint32_t d[3];
    const bool ge = mp_sub<3>(d, r3, z3) == 0;
    uint32_t cy = ge ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      r3[i] = ge ? d[i] : r3[i];
      const uint32_t s = q[i] + cy;
      cy = s < cy ? 1u : 0u;
      q[i] = s;
    }
  }
  t[0] = r3[0];
  t[1] = r3[1];
}

// ---- 192-bit two's-complement helpers
NCG_DI void s192_add_small(uint32_t (&a)[6], int delta, bool on) {  // a += delta (delta = +1 / -1) when `on`
  uint32_t d[6];
#pragma unroll
  for (int i = 0; i < 6; i++) d[i] = delta < 0 ? 0xffffffffu : 0u;
  if (delta > 0) d[0] = 1u;
  uint32_t r[6];
  mp_add<6>(r, a, d);
#pragma unroll
  for (int i = 0; i < 6; i++) a[i] = on ? r[i] : a[i];
}
// a non-negative and a > half ?
NCG_DI bool s192_above(const uint32_t (&a)[6], const uint32_t (&half)[6]) {
  uint32_t d[6];
  const bool gt = mp_sub<6>(d, half, a) != 0;  // half - a borrows  <=>  a > half (unsigned)
  return gt && (a[5] >> 31) == 0;
}
NCG_DI void s192_sub_if(uint32_t (&a)[6], const uint32_t (&m)[6], bool on) {
  uint32_t r[6];
  mp_sub<6>(r, a, m);
#pragma unroll
  for (int i = 0; i < 6; i++) a[i] = on ? r[i] : a[i];
}

// k = k1 + k2 z^2 (mod r), |k1|, |k2| <= z^2/2 + 1.  out[0] = k1, out[1] = k2.
NCG_DI void bls_endo_split2(uint32_t (&out)[2][6], const uint32_t (&k)[8]) {
  uint32_t q[5], t[4];
  bls_divmod_z2(q, t, k);
  uint32_t x6[6], h6[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    x6[i] = i < 4 ? BlsEndo::X2[i] : 0u;
    out[0][i] = i < 4 ? t[i] : 0u;
    out[1][i] = i < 5 ? q[i] : 0u;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) h6[i] = (x6[i] >> 1) | (i < 5 ? (x6[i < 5 ? i + 1 : 5] << 31) : 0u);  // z^2 / 2
  // q > z^2/2:  q z^2 = (q - z^2 + 1) z^2 - 1  (mod r = z^4 - z^2 + 1)
  const bool big2 = s192_above(out[1], h6);
  s192_sub_if(out[1], x6, big2);
  s192_add_small(out[1], +1, big2);
  s192_add_small(out[0], -1, big2);
  // t > z^2/2:  t = (t - z^2) + z^2
  const bool big1 = s192_above(out[0], h6);
  s192_sub_if(out[0], x6, big1);
  s192_add_small(out[1], +1, big1);
}

// k = d0 + d1 z + d2 z^2 + d3 z^3 (mod r), |d_e| <= z/2 + 1
NCG_DI void bls_endo_split4(uint32_t (&out)[4][6], const uint32_t (&k)[8]) {
  uint32_t q[5], t[4];
  bls_divmod_z2(q, t, k);
  uint32_t q4[4] = {q[0], q[1], q[2], q[3]};  // q <= z^2 for k < r
  uint32_t hi[3], lo[2];
  bls_divmod_z(hi, lo, t);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    out[0][i] = i < 2 ? lo[i] : 0u;
    out[1][i] = i < 3 ? hi[i] : 0u;
  }
  bls_divmod_z(hi, lo, q4);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    out[2][i] = i < 2 ? lo[i] : 0u;
    out[3][i] = i < 3 ? hi[i] : 0u;
  }
  uint32_t z6[6] = {BlsEndo::Z[0], BlsEndo::Z[1], 0u, 0u, 0u, 0u};
  uint32_t h6[6] = {(BlsEndo::Z[0] >> 1) | (BlsEndo::Z[1] << 31), BlsEndo::Z[1] >> 1, 0u, 0u, 0u, 0u};
#pragma unroll
  for (int e = 0; e < 3; e++) {
    const bool big = s192_above(out[e], h6);
    s192_sub_if(out[e], z6, big);
    s192_add_small(out[e + 1], +1, big);
  }
  // d3 > z/2:  d3 z^3 = (d3 - z) z^3 + z^4 = (d3 - z) z^3 + z^2 - 1  (mod r)
  const bool big3 = s192_above(out[3], h6);
  s192_sub_if(out[3], z6, big3);
  s192_add_small(out[2], +1, big3);
  s192_add_small(out[0], -1, big3);
}

}  // namespace ncg
