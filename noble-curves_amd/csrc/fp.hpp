// Prime-field arithmetic for gfx950: 32-bit limbs in VGPRs; Montgomery form for bls12-381,
// plain residues with special-form folding for secp256k1 and ed25519 (PR::FOLD).
//
// Reproduces the *values* of the reference's `_Field` ops (src/abstract/modular.ts:940-982:
// add/sub/neg/mul/sqr/inv = canonical residues mod ORDER) - here residues are held as
// a*R mod p (R = 2^(32N)) so that `mul` is one interleaved multiply/reduce (CIOS) instead of
// the reference's BigInt `a*b % p` (modular.ts:956, :50-54).  Conversion to/from canonical
// residues happens only at kernel load/store.
//
// Integer-multiply throughput bounds this code (v_mad_u64_u32); there is no MFMA use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "consts_gen.hpp"

#define NCG_DI __host__ __device__ __forceinline__
#ifndef NCG_MUL_INLINE
#define NCG_MUL_INLINE 0
#endif
#if NCG_MUL_INLINE
#define NCG_MULFN __host__ __device__ __forceinline__
#else
// out-of-line so an EC add (10-16 field muls) stays a few KB of code instead of ~100 KB:
// the CU instruction cache is 64 KB.
#define NCG_MULFN __host__ __device__ __noinline__
#endif

namespace ncg {

template <class PR>
struct Fp {
  static constexpr int N = PR::N;
  using Params = PR;
  uint32_t v[N];

  static NCG_DI Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = 0;
    return r;
  }
  static NCG_DI Fp one() {  // Montgomery form of 1
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = PR::R1[i];
    return r;
  }
  template <class ARR>
  static NCG_DI Fp from_const(const ARR& a) {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = a[i];
    return r;
  }
  NCG_DI bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= v[i];
    return o == 0;
  }
  NCG_DI bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= v[i] ^ b.v[i];
    return o == 0;
  }
  NCG_DI bool operator!=(const Fp& b) const { return !(*this == b); }
};

// r = a - p if a >= p (a < 2p, possibly with an extra carry bit `hi`)
template <class PR>
NCG_DI void fp_cond_sub_p(uint32_t (&a)[PR::N], uint32_t hi) {
  constexpr int N = PR::N;
  uint32_t s[N];
  uint32_t bw = 0;
#pragma unroll
  for (int j = 0; j < N; j++) s[j] = __builtin_subc(a[j], (uint32_t)PR::P[j], bw, &bw);
  bool ge = (hi != 0) || (bw == 0);
#pragma unroll
  for (int j = 0; j < N; j++) a[j] = ge ? s[j] : a[j];
}

template <class PR>
NCG_DI Fp<PR> operator+(const Fp<PR>& a, const Fp<PR>& b) {  // modular.ts:950
  constexpr int N = PR::N;
  Fp<PR> r;
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < N; j++) r.v[j] = __builtin_addc(a.v[j], b.v[j], c, &c);
  fp_cond_sub_p<PR>(r.v, PR::TOP_SPARE ? 0u : c);
  return r;
}

template <class PR>
NCG_DI Fp<PR> operator-(const Fp<PR>& a, const Fp<PR>& b) {  // modular.ts:953
  constexpr int N = PR::N;
  Fp<PR> r;
  uint32_t bw = 0;
#pragma unroll
  for (int j = 0; j < N; j++) r.v[j] = __builtin_subc(a.v[j], b.v[j], bw, &bw);
  // add p back when the subtraction borrowed
  uint32_t m = 0u - bw;
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < N; j++) r.v[j] = __builtin_addc(r.v[j], (uint32_t)PR::P[j] & m, c, &c);
  return r;
}

template <class PR>
NCG_DI Fp<PR> fp_neg(const Fp<PR>& a) {  // modular.ts:940
  constexpr int N = PR::N;
  Fp<PR> r;
  uint32_t bw = 0;
#pragma unroll
  for (int j = 0; j < N; j++) r.v[j] = __builtin_subc((uint32_t)PR::P[j], a.v[j], bw, &bw);
  bool z = a.is_zero();
#pragma unroll
  for (int j = 0; j < N; j++) r.v[j] = z ? 0u : r.v[j];
  return r;
}

template <class PR>
NCG_DI Fp<PR> fp_dbl(const Fp<PR>& a) {
  return a + a;
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(NCG_NO_ASM_PRODUCT)
// acc (96 bits: lo64 + ex) += a*b.  v_mad_u64_u32 adds the 64-bit product into lo64 and reports
// the carry in VCC; one v_addc folds it into `ex`.  Two instructions per partial product instead of
// three (mad + two carry adds), and no SGPR-carried chains (which cost s_nop hazard padding).
__device__ __forceinline__ void mac96(uint64_t& lo64, uint32_t& ex, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(lo64), "+v"(ex)
      : "v"(a), "v"(b)
      : "vcc");
}
#endif

#if defined(__HIP_DEVICE_COMPILE__) && !defined(NCG_NO_ASM_PRODUCT)
// Montgomery product by columns (finely integrated product scanning) on the 96-bit accumulator:
// 2N^2 + N multiplies, one carry instruction per multiply.
template <class PR>
__device__ __forceinline__ void fp_mul_fips_asm(uint32_t (&r)[PR::N], const uint32_t (&a)[PR::N],
                                                const uint32_t (&b)[PR::N]) {
  constexpr int N = PR::N;
  uint32_t q[N];
  uint64_t lo = 0;
  uint32_t ex = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) mac96(lo, ex, a[i], b[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) mac96(lo, ex, q[i], (uint32_t)PR::P[k - i]);
    q[k] = (uint32_t)lo * PR::INV;
    mac96(lo, ex, q[k], (uint32_t)PR::P[0]);
    lo = (lo >> 32) | ((uint64_t)ex << 32);
    ex = 0;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = k - N + 1; i < N; i++) mac96(lo, ex, a[i], b[k - i]);
#pragma unroll
    for (int i = k - N + 1; i < N; i++) mac96(lo, ex, q[i], (uint32_t)PR::P[k - i]);
    r[k - N] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)ex << 32);
    ex = 0;
  }
  r[N - 1] = (uint32_t)lo;
  fp_cond_sub_p<PR>(r, (uint32_t)(lo >> 32));
}
#endif

// Montgomery product a*b*R^-1 mod p, CIOS with two carry chains per row
// (lo-half chain and hi-half chain of the N partial products).
template <class PR>
NCG_DI void fp_mul_body(uint32_t (&r)[PR::N], const uint32_t (&a)[PR::N], const uint32_t (&b)[PR::N]) {
  constexpr int N = PR::N;
  uint32_t T[N + 2];
#pragma unroll
  for (int i = 0; i < N + 2; i++) T[i] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t lo[N], hi[N];
    uint32_t c;
#pragma unroll
    for (int j = 0; j < N; j++) {
      uint64_t t = (uint64_t)a[j] * b[i];
      lo[j] = (uint32_t)t;
      hi[j] = (uint32_t)(t >> 32);
    }
    c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) T[j] = __builtin_addc(T[j], lo[j], c, &c);
    T[N] = __builtin_addc(T[N], 0u, c, &c);
    T[N + 1] += c;
    c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) T[j + 1] = __builtin_addc(T[j + 1], hi[j], c, &c);
    T[N + 1] += c;
    uint32_t m = T[0] * PR::INV;
#pragma unroll
    for (int j = 0; j < N; j++) {
      uint64_t t = (uint64_t)m * (uint32_t)PR::P[j];
      lo[j] = (uint32_t)t;
      hi[j] = (uint32_t)(t >> 32);
    }
    c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) T[j] = __builtin_addc(T[j], lo[j], c, &c);
    T[N] = __builtin_addc(T[N], 0u, c, &c);
    T[N + 1] += c;
    c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) T[j + 1] = __builtin_addc(T[j + 1], hi[j], c, &c);
    T[N + 1] += c;
#pragma unroll
    for (int j = 0; j < N + 1; j++) T[j] = T[j + 1];
    T[N + 1] = 0;
  }
#pragma unroll
  for (int j = 0; j < N; j++) r[j] = T[j];
  fp_cond_sub_p<PR>(r, T[N]);
}

// ---- special-form primes p = 2^256 - c (secp256k1: c = 2^32 + 977) or 2p = 2^256 - c
// (ed25519: 2^256 = 2p + 38): plain (non-Montgomery) residues, product folded with 2^256 = c.
// Half the multiply-adds of the Montgomery path (72 instead of 136 v_mad_u64_u32) and no q*p
// carry chains.  Params provide FOLD_LO (low 32 bits of c), FOLD_HI (c >> 32, 0 or 1) and
// FINAL_SUBS (conditional subtractions of p that make the result canonical: 1 or 2).
// r = T mod p for a 512-bit T (16 limbs), folding 2^256 = c twice, then canonical.
template <class PR>
NCG_DI void fp_fold_reduce(uint32_t (&r)[8], const uint32_t (&T)[16]) {
  constexpr int N = 8;
  // first fold: U = L + H*c, c = FOLD_HI*2^32 + FOLD_LO ; U has 8 limbs + a small overflow limb
  uint32_t U[N];
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    uint64_t acc = (uint64_t)T[N + j] * PR::FOLD_LO + T[j];
    acc += carry;
    if (PR::FOLD_HI && j > 0) acc += T[N + j - 1];
    U[j] = (uint32_t)acc;
    carry = acc >> 32;
  }
  if (PR::FOLD_HI) carry += T[2 * N - 1];  // overflow limb < 2^34
  // second fold of the overflow limb (needs up to 3 low limbs), then ripple
  {
    uint64_t acc = (uint64_t)(uint32_t)carry * PR::FOLD_LO + U[0];
    uint64_t hi2 = (carry >> 32) * (uint64_t)PR::FOLD_LO;  // carry may have 34 bits
    U[0] = (uint32_t)acc;
    acc = (acc >> 32) + U[1] + (uint32_t)hi2;
    if (PR::FOLD_HI) acc += (uint32_t)carry;
    U[1] = (uint32_t)acc;
    acc = (acc >> 32) + U[2] + (hi2 >> 32);
    if (PR::FOLD_HI) acc += (carry >> 32);
    U[2] = (uint32_t)acc;
    uint32_t c = (uint32_t)(acc >> 32);
#pragma unroll
    for (int j = 3; j < N; j++) U[j] = __builtin_addc(U[j], 0u, c, &c);
    // a carry out of limb 7 means the value wrapped 2^256 once more: add c again (cannot recur)
    uint32_t m = 0u - c;
    uint32_t c2 = 0;
    U[0] = __builtin_addc(U[0], (uint32_t)PR::FOLD_LO & m, 0u, &c2);
    U[1] = __builtin_addc(U[1], (PR::FOLD_HI ? 1u : 0u) & m, c2, &c2);
#pragma unroll
    for (int j = 2; j < N; j++) U[j] = __builtin_addc(U[j], 0u, c2, &c2);
  }
#pragma unroll
  for (int j = 0; j < N; j++) r[j] = U[j];
#pragma unroll
  for (int k = 0; k < PR::FINAL_SUBS; k++) fp_cond_sub_p<PR>(r, 0u);
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(NCG_NO_ASM_PRODUCT)
// 512-bit product by columns (product scanning): T[k] = low word of column k after carries.
template <int N>
__device__ __forceinline__ void mul_columns_asm(uint32_t (&T)[2 * N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {
  uint64_t lo = 0;
  uint32_t ex = 0;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int j = k - i;
      if (j >= 0 && j < N) mac96(lo, ex, a[i], b[j]);
    }
    T[k] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)ex << 32);
    ex = 0;
  }
  T[2 * N - 1] = (uint32_t)lo;
}
// Square by columns: off-diagonal products once, the 96-bit column sum doubled, then the
// diagonal term and the carry from the previous column.
template <int N>
__device__ __forceinline__ void sqr_columns_asm(uint32_t (&T)[2 * N], const uint32_t (&a)[N]) {
  uint64_t carry = 0;  // (column sum) >> 32 of the previous column, below 2^64
#pragma unroll
  for (int k = 0; k < 2 * N - 1; k++) {
    uint64_t lo = 0;
    uint32_t ex = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int j = k - i;
      if (j > i && j < N) mac96(lo, ex, a[i], a[j]);
    }
    // double the off-diagonal sum (below 2^67, so nothing is shifted out of ex)
    ex = (ex << 1) | (uint32_t)(lo >> 63);
    lo <<= 1;
    if ((k & 1) == 0) mac96(lo, ex, a[k / 2], a[k / 2]);
    uint64_t nl = lo + carry;
    ex += nl < lo ? 1u : 0u;
    T[k] = (uint32_t)nl;
    carry = (nl >> 32) | ((uint64_t)ex << 32);
  }
  T[2 * N - 1] = (uint32_t)carry;
}
#endif

template <class PR>
NCG_DI void fp_mul_fold_body(uint32_t (&r)[PR::N], const uint32_t (&a)[PR::N], const uint32_t (&b)[PR::N]) {
  static_assert(PR::N == 8, "fold path is for 256-bit special primes");
  constexpr int N = 8;
  uint32_t T[2 * N];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NCG_NO_ASM_PRODUCT)
  mul_columns_asm<N>(T, a, b);
#else
#pragma unroll
  for (int i = 0; i < 2 * N; i++) T[i] = 0;
  // 512-bit product, two carry chains per row
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t lo[N], hi[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
      uint64_t t = (uint64_t)a[j] * b[i];
      lo[j] = (uint32_t)t;
      hi[j] = (uint32_t)(t >> 32);
    }
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) T[i + j] = __builtin_addc(T[i + j], lo[j], c, &c);
    T[i + N] += c;  // cannot overflow: partial sums are below 2^(32(i+9))
    c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) T[i + j + 1] = __builtin_addc(T[i + j + 1], hi[j], c, &c);
    if (i + N + 1 < 2 * N) T[i + N + 1] += c;
  }
#endif
  fp_fold_reduce<PR>(r, T);
}

// square: the 28 off-diagonal products once, doubled, plus the 8 squares (36 instead of 64
// multiply-adds)
template <class PR>
NCG_DI void fp_sqr_fold_body(uint32_t (&r)[PR::N], const uint32_t (&a)[PR::N]) {
  static_assert(PR::N == 8, "fold path is for 256-bit special primes");
  constexpr int N = 8;
  uint32_t T[2 * N];
#if defined(__HIP_DEVICE_COMPILE__) && defined(NCG_ASM_SQR)
#if NCG_ASM_SQR == 2
  mul_columns_asm<N>(T, a, a);
#else
  sqr_columns_asm<N>(T, a);
#endif
  fp_fold_reduce<PR>(r, T);
  return;
#endif
#pragma unroll
  for (int i = 0; i < 2 * N; i++) T[i] = 0;
#pragma unroll
  for (int i = 0; i < N - 1; i++) {
    uint32_t lo[N], hi[N];
#pragma unroll
    for (int j = i + 1; j < N; j++) {
      uint64_t t = (uint64_t)a[j] * a[i];
      lo[j] = (uint32_t)t;
      hi[j] = (uint32_t)(t >> 32);
    }
    uint32_t c = 0;
#pragma unroll
    for (int j = i + 1; j < N; j++) T[i + j] = __builtin_addc(T[i + j], lo[j], c, &c);
    T[i + N] += c;
    c = 0;
#pragma unroll
    for (int j = i + 1; j < N; j++) T[i + j + 1] = __builtin_addc(T[i + j + 1], hi[j], c, &c);
    if (i + N + 1 < 2 * N) T[i + N + 1] += c;
  }
  // double (the off-diagonal sum is below 2^511)
#pragma unroll
  for (int k = 2 * N - 1; k > 0; k--) T[k] = (T[k] << 1) | (T[k - 1] >> 31);
  T[0] <<= 1;
  // add the squares a_i^2 at limb 2i
  {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t t = (uint64_t)a[i] * a[i];
      T[2 * i] = __builtin_addc(T[2 * i], (uint32_t)t, c, &c);
      T[2 * i + 1] = __builtin_addc(T[2 * i + 1], (uint32_t)(t >> 32), c, &c);
    }
  }
  fp_fold_reduce<PR>(r, T);
}

template <class PR>
NCG_MULFN Fp<PR> fp_mul(Fp<PR> a, Fp<PR> b) {  // modular.ts:956
  Fp<PR> r;
  if constexpr (PR::FOLD) {
    fp_mul_fold_body<PR>(r.v, a.v, b.v);
  } else {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NCG_NO_ASM_PRODUCT)
    fp_mul_fips_asm<PR>(r.v, a.v, b.v);
#else
    fp_mul_body<PR>(r.v, a.v, b.v);
#endif
  }
  return r;
}

template <class PR>
NCG_MULFN Fp<PR> fp_sqr(Fp<PR> a) {  // modular.ts:947
  Fp<PR> r;
  if constexpr (PR::FOLD) {
    fp_sqr_fold_body<PR>(r.v, a.v);
  } else {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NCG_NO_ASM_PRODUCT)
    fp_mul_fips_asm<PR>(r.v, a.v, a.v);
#else
    fp_mul_body<PR>(r.v, a.v, a.v);
#endif
  }
  return r;
}

template <class PR>
NCG_DI Fp<PR> operator*(const Fp<PR>& a, const Fp<PR>& b) {
  return fp_mul<PR>(a, b);
}

// canonical residue (little-endian 32-bit limbs) -> Montgomery form
template <class PR>
NCG_DI Fp<PR> fp_to_mont(const Fp<PR>& a) {
  return fp_mul<PR>(a, Fp<PR>::from_const(PR::R2));
}
// Montgomery form -> canonical residue in [0, p)
template <class PR>
NCG_DI Fp<PR> fp_from_mont(const Fp<PR>& a) {
  Fp<PR> o = Fp<PR>::zero();
  o.v[0] = 1;
  return fp_mul<PR>(a, o);
}

// a^(2^n)
template <class PR>
NCG_DI Fp<PR> fp_sqr_n(Fp<PR> a, int n) {
  for (int i = 0; i < n; i++) a = fp_sqr<PR>(a);
  return a;
}

// Fermat inversion a^(p-2) (square-and-multiply over the constant exponent, MSB first);
// same value as the reference's Euclidean invert() (modular.ts:159-182); 0 -> 0.
template <class PR>
NCG_DI Fp<PR> fp_inv(const Fp<PR>& a) {
  Fp<PR> r = Fp<PR>::one();
  bool started = false;
  for (int w = PR::N - 1; w >= 0; w--) {
    uint32_t word = PR::P[w];
    if (w == 0) word -= 2u;  // p[0] >= 2 for every field here
    for (int bit = 31; bit >= 0; bit--) {
      if (started) r = fp_sqr<PR>(r);
      if ((word >> bit) & 1u) {
        r = started ? fp_mul<PR>(r, a) : a;
        started = true;
      }
    }
  }
  return r;
}

// load/store of canonical residues in wire format (little-endian bytes == LE 32-bit limbs)
template <class PR>
NCG_DI Fp<PR> fp_load(const uint32_t* __restrict__ p) {
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < PR::N; i++) r.v[i] = p[i];
  return r;
}
template <class PR>
NCG_DI void fp_store(uint32_t* __restrict__ p, const Fp<PR>& a) {
#pragma unroll
  for (int i = 0; i < PR::N; i++) p[i] = a.v[i];
}

using FpSecp = Fp<ParamsSecpP>;
using FpEd = Fp<ParamsEdP>;
using FpBls = Fp<ParamsBlsP>;

}  // namespace ncg
