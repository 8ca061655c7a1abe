// bls12-381 scalar field Fr in radix 2^29 for the NTT butterflies (ntt.hip): 9 limbs, Montgomery
// products with R = 2^261, lazy additions.
//
// Values of the reference's Fr ops (src/abstract/modular.ts:940-982 on bls12_381_Fr, used by FFTCore,
// src/abstract/fft.ts:454-478) are reproduced at the pass boundaries only; inside a pass an element is
// 9 limbs of up to 32 bits whose value is only CONGRUENT to the residue.
//
// Why this form for the butterflies (cycle figures: tools/valu_rates.hip on MI355X): the 8 x 32-bit
// Montgomery product of fp.hpp is 136 v_mad_u64_u32 each followed by a v_addc (the 96-bit column),
// then a conditional subtraction, and the butterfly's + and - are carry chains with a conditional
// subtraction each - about 1 700 issue cycles per butterfly.  With 29-bit limbs
//   * a column of 9 + 8 products fits 64 bits: 81 + 72 multiply-adds with no carry instruction;
//   * r = 1 (mod 2^29): the Montgomery quotient digit is the negated low limb (no multiply) and
//     q * r[0] is q itself;
//   * a + t is 9 adds, a - t is 9 (add-bias, sub) pairs, the limbs are brought back below 2^29 + 8
//     only every third stage and the value below 2^256 once per pass.
// About 1 200 issue cycles per butterfly.
//
// Bounds (checked by tests/test_host_logic.py on the host twin with every operand at its maximum):
//   * "limb bound A": every limb below A * 2^29 (limb 8 may use the full 32 bits).
//   * fr29_mont(b, w): b with limb bound <= 6, w with exact limbs (< 2^29): columns stay below
//     (9*6 + 8) * 2^58 + 2^36 < 2^64.  Result: exact limbs, value below val(b) * r / 2^261 + r.
//   * a pass starts from values below 2^256 (2.21 r), every stage adds at most 3 r (the bias): after 10
//     stages below 33 r < 2^260, far inside limb 8's 32 bits (value < 2^264).
#pragma once
#include "fe9.hpp"
#include "fr29_asm_gen.hpp"

// 1: the multiply-add chains of fr29_mont as one asm block per column (tools/gen_fr29_asm.py); 0: one statement per instruction
#ifndef NCG_FR29_BLOCKS
#define NCG_FR29_BLOCKS 1
#endif

namespace ncg {

struct Fr29 {
  uint32_t v[9];
};

// host twin only: counts 64-bit column / 32-bit limb overflows so the unit tests can assert there are none
inline int& fr29_overflows() {
  static int n = 0;
  return n;
}

NCG_DI void fr29_mac(uint64_t& acc, uint32_t a, uint32_t b) {
#ifdef __HIP_DEVICE_COMPILE__
  fe9_mac(acc, a, b);
#else
  const uint64_t p = (uint64_t)a * b, s = acc + p;
  if (s < acc) fr29_overflows()++;
  acc = s;
#endif
}
NCG_DI void fr29_mac_k(uint64_t& acc, uint32_t a, uint32_t k) {
#ifdef __HIP_DEVICE_COMPILE__
  fe9_mac_k(acc, a, k);
#else
  fr29_mac(acc, a, k);
#endif
}
NCG_DI uint32_t fr29_add32(uint32_t a, uint32_t b) {
#if !defined(__HIP_DEVICE_COMPILE__)
  if ((uint64_t)a + b > 0xFFFFFFFFull) fr29_overflows()++;
#endif
  return a + b;
}

// (hi:lo) >> sh, low 32 bits (one v_alignbit_b32)
NCG_DI uint32_t fr29_funnel(uint32_t hi, uint32_t lo, int sh) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}
// 8 x 32-bit little-endian words (a value below 2^256) -> exact limbs
NCG_DI Fr29 fr29_from_words(const uint32_t (&w)[8]) {
  Fr29 r;
  r.v[0] = w[0] & FE9_MASK;
  r.v[1] = fr29_funnel(w[1], w[0], 29) & FE9_MASK;
  r.v[2] = fr29_funnel(w[2], w[1], 26) & FE9_MASK;
  r.v[3] = fr29_funnel(w[3], w[2], 23) & FE9_MASK;
  r.v[4] = fr29_funnel(w[4], w[3], 20) & FE9_MASK;
  r.v[5] = fr29_funnel(w[5], w[4], 17) & FE9_MASK;
  r.v[6] = fr29_funnel(w[6], w[5], 14) & FE9_MASK;
  r.v[7] = fr29_funnel(w[7], w[6], 11) & FE9_MASK;
  r.v[8] = w[7] >> 8;  // bits 232..255
  return r;
}
// exact limbs of a value below 2^256 -> 8 words
NCG_DI void fr29_to_words(uint32_t (&w)[8], const Fr29& a) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int bit = 32 * k, limb = bit / 29, sh = bit % 29;
    uint64_t acc = (uint64_t)a.v[limb] >> sh;
    acc |= (uint64_t)a.v[limb + 1] << (29 - sh);
    if (limb + 2 < 9) acc |= (uint64_t)a.v[limb + 2] << (58 - sh);
    w[k] = (uint32_t)acc;
  }
}

// Montgomery product b * w / 2^261 (mod r).  b: limb bound <= 6; w: exact limbs.  Exact limbs out.
NCG_DI Fr29 fr29_mont(const Fr29& b, const Fr29& w) {
#if defined(__HIP_DEVICE_COMPILE__) && NCG_FR29_BLOCKS
  {
    Fr29 o;
    NCG_FR29_MONT_BLOCKS(b, w, o)
    return o;
  }
#endif
  uint32_t q[9];
  Fr29 o;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) fr29_mac(acc, b.v[i], w.v[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) fr29_mac_k(acc, q[i], Fr29PR::P[k - i]);
    q[k] = (0u - (uint32_t)acc) & FE9_MASK;  // -r^-1 = -1 (mod 2^29)
    fr29_mac_k(acc, q[k], 1u);               // r[0] = 1: the low 29 bits cancel
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) fr29_mac(acc, b.v[i], w.v[k - i]);
#pragma unroll
    for (int i = k - 8; i < 9; i++) fr29_mac_k(acc, q[i], Fr29PR::P[k - i]);
    o.v[k - 9] = (uint32_t)acc & FE9_MASK;
    acc >>= 29;
  }
  o.v[8] = (uint32_t)acc;
  return o;
}

NCG_DI Fr29 fr29_add(const Fr29& a, const Fr29& t) {
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = fr29_add32(a.v[i], t.v[i]);
  return r;
}
// a - t as a + (3 r - t): t must have exact limbs and limb 8 at most BIAS[8] (any value below 2.99 r; the
// products that come here are below 1.5 r).  One v_sad_u32 per limb
// (|BIAS - t| + a with BIAS >= t) instead of a subtract and an add.
NCG_DI Fr29 fr29_sub(const Fr29& a, const Fr29& t) {
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
#ifdef __HIP_DEVICE_COMPILE__
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r.v[i]) : "s"(Fr29PR::BIAS[i]), "v"(t.v[i]), "v"(a.v[i]));
#else
    if (t.v[i] > Fr29PR::BIAS[i]) fr29_overflows()++;
    r.v[i] = fr29_add32(a.v[i], Fr29PR::BIAS[i] - t.v[i]);
#endif
  }
  return r;
}

// weak normalisation: limbs 0..7 back below 2^29 + 8 (limb 8 collects the carry), value unchanged
NCG_DI Fr29 fr29_weak(const Fr29& a) {
  Fr29 r;
  r.v[0] = a.v[0] & FE9_MASK;
#pragma unroll
  for (int i = 1; i < 8; i++) r.v[i] = (a.v[i] & FE9_MASK) + (a.v[i - 1] >> 29);
  r.v[8] = fr29_add32(a.v[8], a.v[7] >> 29);
  return r;
}

// one fold of the bits at and above 2^255 (2^255 = C255 mod r) with an exact carry chain:
// value < 2^264 in, exact limbs and value < 2^255 * (1 + 0.0944 * h) out, h = value >> 255 (before the fold)
NCG_DI Fr29 fr29_fold255(const Fr29& a) {
  // first make limb 8 exact enough to read h: propagate the carries (limbs up to 32 bits)
  uint32_t t[9];
  uint32_t cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t e = (uint64_t)a.v[i] + cy;
    t[i] = (uint32_t)e & FE9_MASK;
    cy = (uint32_t)(e >> 29);
  }
  t[8] = fr29_add32(a.v[8], cy);
  const uint32_t h = t[8] >> 23;
  t[8] &= (1u << 23) - 1u;
  Fr29 r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    fr29_mac_k(c, h, Fr29PR::C255[i]);
    c += t[i];
    r.v[i] = (uint32_t)c & FE9_MASK;
    c >>= 29;
  }
  fr29_mac_k(c, h, Fr29PR::C255[8]);
  c += t[8];
  r.v[8] = (uint32_t)c;
  return r;
}
// value below 33 r (any limb bound that fits) -> exact limbs, value below 1.29 * 2^255 < 2^256
NCG_DI Fr29 fr29_reduce256(const Fr29& a) { return fr29_fold255(fr29_fold255(a)); }

// exact limbs, value below 2 r -> canonical residue
NCG_DI Fr29 fr29_cond_sub(const Fr29& a) {
  uint32_t s[9];
  int32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int32_t d = (int32_t)a.v[i] - (int32_t)Fr29PR::P[i] + bw;
    s[i] = (uint32_t)d & FE9_MASK;
    bw = d >> 29;
  }
  const int32_t d8 = (int32_t)a.v[8] - (int32_t)Fr29PR::P[8] + bw;
  const bool ge = d8 >= 0;
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = ge ? s[i] : a.v[i];
  r.v[8] = ge ? (uint32_t)d8 : a.v[8];
  return r;
}

}  // namespace ncg
