// Batch point decoding (decompression + validity), SURVEY 8(f) row 1: the step on the input
// side of every batch op (wire bytes -> affine point).
//
//   secp256k1  SEC1 compressed, 33 bytes: Point.fromBytes -> pointFromBytes
//              (src/abstract/weierstrass.ts:566-605) + sqrtMod (src/secp256k1.ts:73-95);
//              assertValidity (:748-766) holds by construction (h = 1)
//   bls12-381 G1 compressed, 48 bytes: coder.decode (src/bls12-381.ts:377-433, flags :436-459,
//              sortBit :346-351) + assertValidity incl. the subgroup check isTorsionFree
//              [x^2]P == phi(P) (src/bls12-381.ts:567-577)
//   bls12-381 G2 compressed, 96 bytes (c1 || c0, src/bls12-381.ts:354-368): the same coder with
//              Fp2.sqrt (src/abstract/tower.ts:476-500) and the psi subgroup check
//              [-x]P == psi(P) (src/bls12-381.ts:599-601, psi: src/abstract/tower.ts:240-247)
//   ed25519    32 bytes: Point.fromBytes(bytes, zip215) (src/abstract/edwards.ts:405-436)
// out_ok[i] = 0 exactly where the reference throws; the affine output is then (0,0).
#include <cstdlib>
#include "knobs.hpp"
#include <vector>

#include "bls_lanes.hpp"
#include "host_api.hpp"

namespace ncg {

// ---------------------------------------------------------------------------- secp256k1
// y = sqrt(v) = v^((p+1)/4) by the reference's addition chain (secp256k1.ts:73-95)
NCG_DI FpSecp secp_sqrt_candidate(const FpSecp& y) {
  using PR = ParamsSecpP;
  FpSecp b2 = fp_sqr<PR>(y) * y;
  FpSecp b3 = fp_sqr<PR>(b2) * y;
  FpSecp b6 = fp_sqr_n<PR>(b3, 3) * b3;
  FpSecp b9 = fp_sqr_n<PR>(b6, 3) * b3;
  FpSecp b11 = fp_sqr_n<PR>(b9, 2) * b2;
  FpSecp b22 = fp_sqr_n<PR>(b11, 11) * b11;
  FpSecp b44 = fp_sqr_n<PR>(b22, 22) * b22;
  FpSecp b88 = fp_sqr_n<PR>(b44, 44) * b44;
  FpSecp b176 = fp_sqr_n<PR>(b88, 88) * b88;
  FpSecp b220 = fp_sqr_n<PR>(b176, 44) * b44;
  FpSecp b223 = fp_sqr_n<PR>(b220, 3) * b3;
  FpSecp t1 = fp_sqr_n<PR>(b223, 23) * b22;
  FpSecp t2 = fp_sqr_n<PR>(t1, 6) * b2;
  return fp_sqr_n<PR>(t2, 2);
}

// in: 33 bytes (02/03 || x big-endian); out: x || y wire (LE limbs)
NCG_DI bool secp_decode_lane(const uint8_t* __restrict__ in, uint32_t* __restrict__ out) {
  using PR = ParamsSecpP;
  const uint8_t head = in[0];
  FpSecp x;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint8_t* b = in + 1 + (7 - i) * 4;
    x.v[i] = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3];
  }
  bool ok = head == 2 || head == 3;
  {  // Fp.isValid(x): x < p
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) (void)__builtin_subc(x.v[i], (uint32_t)PR::P[i], bw, &bw);
    ok = ok && bw != 0;
  }
  FpSecp seven = FpSecp::zero();
  seven.v[0] = 7;
  FpSecp y2 = fp_sqr<PR>(x) * x + seven;  // weierstrassEquation, a = 0, b = 7
  FpSecp y = secp_sqrt_candidate(y2);
  ok = ok && (fp_sqr<PR>(y) == y2);       // "Cannot find square root"
  if (((head & 1u) != 0) != ((y.v[0] & 1u) != 0)) y = fp_neg<PR>(y);
  if (!ok) {
    x = FpSecp::zero();
    y = FpSecp::zero();
  }
  fp_store<PR>(out, x);
  fp_store<PR>(out + 8, y);
  return ok;
}

// ---------------------------------------------------------------------------- bls12-381 G1
// in: 48 bytes; out: x || y wire (12 + 12 LE limbs).  *inf set for the canonical infinity encoding.
NCG_DI bool g1_decode_lane(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, uint8_t* inf) {
  using F = FeBls;
  const uint8_t mask = in[0] & 0xE0;
  const bool compressed = (mask & 0x80) != 0, infinity = (mask & 0x40) != 0, sort = (mask & 0x20) != 0;
  // validateMask (:439-446) and the 48-byte length rule
  bool ok = compressed && !(infinity && sort);
  uint32_t xw[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    const uint8_t* b = in + (11 - i) * 4;
    xw[i] = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3];
  }
  xw[11] &= 0x1FFFFFFFu;  // clear the three flag bits
  uint32_t any = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) any |= xw[i];
  *inf = 0;
  if (infinity) {  // every payload byte must be zero
    ok = ok && any == 0;
    for (int i = 0; i < 24; i++) out[i] = 0;
    *inf = ok ? 1 : 0;
    return ok;
  }
  {  // Fp.fromBytes: x < p
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) (void)__builtin_subc(xw[i], (uint32_t)BlsFpConsts::P32[i], bw, &bw);
    ok = ok && bw != 0;
  }
  F x = fe29_from_wire(xw);
  Fe29<1> four;
#pragma unroll
  for (int i = 0; i < 14; i++) four.v[i] = ParamsBls29::FOUR[i];
  Fe29<2> rhs = (f_sqr(x) * x + four) * Fe29<1>::one();  // x^3 + 4, bound back to 2
  Fe29<2> y = fe29_pow_sqrt(rhs);
  ok = ok && f_eq(f_sqr(y), rhs);  // Fp.sqrt throws when there is no root
  // sort bit: canonical y > (p-1)/2
  uint32_t yw[12];
  fe29_to_wire(yw, y);
  bool big;
  {
    uint32_t bw = 0;  // HALF_P - y borrows  <=>  y > HALF_P
#pragma unroll
    for (int i = 0; i < 12; i++) (void)__builtin_subc((uint32_t)BlsFpConsts::HALF_P[i], yw[i], bw, &bw);
    big = bw != 0;
  }
  if (big != sort) {  // y = p - y  (y != 0: the curve has no point of order 2)
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) yw[i] = __builtin_subc((uint32_t)BlsFpConsts::P32[i], yw[i], bw, &bw);
    y = f_neg(y) * Fe29<1>::one();
  }
  // assertValidity: on the curve by construction; subgroup: [x]([x]P) negated twice == phi(P)
  {
    Jac<F> P{x, y, F::one()};
    Jac<F> xP = jac_neg(bls_mul_by_x(P));
    Jac<F> u2P = bls_mul_by_x(xP);
    Fe29<1> beta;
#pragma unroll
    for (int i = 0; i < 14; i++) beta.v[i] = ParamsBls29::G1_BETA[i];
    // u2P (X, Y, Z) == (beta*x, y) affine  <=>  X == beta*x*Z^2, Y == y*Z^3, Z != 0
    auto zz = f_sqr(u2P.Z);
    bool same = !f_eqz(u2P.Z) && f_eq(u2P.X, x * beta * zz) && f_eq(u2P.Y, y * zz * u2P.Z);
    ok = ok && same;
  }
#pragma unroll
  for (int i = 0; i < 12; i++) {
    out[i] = ok ? xw[i] : 0u;
    out[12 + i] = ok ? yw[i] : 0u;
  }
  return ok;
}

// ---------------------------------------------------------------------------- bls12-381 G2
// in: 96 bytes (x.c1 || x.c0, big-endian, flags in byte 0); out: x.c0 x.c1 y.c0 y.c1 wire
// SUBGROUP = false leaves out the psi test (the split device pipeline runs it lane-paired in a second kernel)
template <bool SUBGROUP = true>
NCG_DI bool g2_decode_lane(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, uint8_t* inf) {
  using F = FeBls2;
  const uint8_t mask = in[0] & 0xE0;
  const bool compressed = (mask & 0x80) != 0, infinity = (mask & 0x40) != 0, sort = (mask & 0x20) != 0;
  bool ok = compressed && !(infinity && sort);
  uint32_t x0w[12], x1w[12];
  be48_to_words(in, x1w);
  be48_to_words(in + 48, x0w);
  x1w[11] &= 0x1FFFFFFFu;
  uint32_t any = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) any |= x0w[i] | x1w[i];
  *inf = 0;
  if (infinity) {
    ok = ok && any == 0;
    for (int i = 0; i < 48; i++) out[i] = 0;
    *inf = ok ? 1 : 0;
    return ok;
  }
  ok = ok && words12_lt_p(x0w) && words12_lt_p(x1w);  // decodeFp -> Fp.fromBytes range rule
  Fe29x2<2> x{fe29_from_wire(x0w), fe29_from_wire(x1w)};
  const Fe29<1> four = fe29_const(ParamsBls29::FOUR);
  auto x3 = f_sqr(x) * x;  // b = 4 (1 + u), src/bls12-381.ts:321-345
  Fe29x2<2> rhs{(x3.c0 + four) * Fe29<1>::one(), (x3.c1 + four) * Fe29<1>::one()};
  Fe29x2<2> y;
  ok = ok && fe29x2_sqrt(rhs, y);
  uint32_t y0w[12], y1w[12];
  fe29_to_wire(y0w, y.c0);
  fe29_to_wire(y1w, y.c1);
  // sortBit over [c1, c0] (:346-351, :476-479): the first non-zero part decides
  uint32_t any1 = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) any1 |= y1w[i];
  const bool big = any1 ? words12_gt_half_p(y1w) : words12_gt_half_p(y0w);
  if (big != sort) {
    words12_neg_mod_p(y0w);
    words12_neg_mod_p(y1w);
    auto ny = f_neg(y);
    y = {ny.c0 * Fe29<1>::one(), ny.c1 * Fe29<1>::one()};
  }
  if constexpr (SUBGROUP) {  // subgroup: -[|x|]P == psi(P), psi(x, y) = (conj(x) PSI_X, conj(y) PSI_Y)
    Jac<F> P{x, y, F::one()};
    Jac<F> xP = jac_neg(bls_mul_by_x(P));
    const Fe29x2<1> psx{fe29_const(ParamsBls29::PSI_X_C0), fe29_const(ParamsBls29::PSI_X_C1)};
    const Fe29x2<1> psy{fe29_const(ParamsBls29::PSI_Y_C0), fe29_const(ParamsBls29::PSI_Y_C1)};
    Fe29x2<4> cx{x.c0, f_neg(x.c1)}, cy{y.c0, f_neg(y.c1)};
    auto px = cx * psx;
    auto py = cy * psy;
    auto zz = f_sqr(xP.Z);
    auto dx = xP.X - px * zz;
    auto dy = xP.Y - py * zz * xP.Z;
    ok = ok && !f_eqz(xP.Z) && f_eqz(dx) && f_eqz(dy);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) {
    out[i] = ok ? x0w[i] : 0u;
    out[12 + i] = ok ? x1w[i] : 0u;
    out[24 + i] = ok ? y0w[i] : 0u;
    out[36 + i] = ok ? y1w[i] : 0u;
  }
  return ok;
}

// ---------------------------------------------------------------------------- ed25519
NCG_DI bool ed_decode_lane(const uint8_t* __restrict__ in, bool zip215, uint32_t* __restrict__ out) {
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint8_t* b = in + 4 * i;
    w[i] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
  }
  FEd x, y;
  bool ok = ed_decompress(w, zip215, x, y);
  if (!ok) {
    x = FEd::zero();
    y = FEd::zero();
  }
  FieldWire<FEd>::store(out, x);
  FieldWire<FEd>::store(out + 8, y);
  return ok;
}

// ---------------------------------------------------------------------------- kernels
__global__ void __launch_bounds__(256) k_decode_secp(const uint8_t* __restrict__ in, uint32_t* __restrict__ out,
                                                     uint8_t* __restrict__ ok, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ok[i] = secp_decode_lane(in + (size_t)i * 33, out + (size_t)i * 16) ? 1 : 0;
}
#ifndef NCG_DEC_G1_MINW
#define NCG_DEC_G1_MINW 2
#endif
#ifndef NCG_DEC_G2_MINW
#define NCG_DEC_G2_MINW 2
#endif
__global__ void __launch_bounds__(256, NCG_DEC_G1_MINW) k_decode_g1(const uint8_t* __restrict__ in, uint32_t* __restrict__ out,
                                                   uint8_t* __restrict__ ok, uint8_t* __restrict__ inf, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t f = 0;
  ok[i] = g1_decode_lane(in + (size_t)i * 48, out + (size_t)i * 24, &f) ? 1 : 0;
  inf[i] = f;
}
#ifdef NCG_AB_BUILD  // one lane per point, decompression + subgroup test fused: 587 spilled registers; replaced by the two stages below
__global__ void __launch_bounds__(128, NCG_DEC_G2_MINW) k_decode_g2(const uint8_t* __restrict__ in, uint32_t* __restrict__ out,
                                                   uint8_t* __restrict__ ok, uint8_t* __restrict__ inf, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t f = 0;
  ok[i] = g2_decode_lane(in + (size_t)i * 96, out + (size_t)i * 48, &f) ? 1 : 0;
  inf[i] = f;
}
#endif
// ---- split G2 decoder (device default): stage A decompresses (range rules, Fp2 square root, sort bit) one
// point per lane in the unpaired form; stage B runs the subgroup test -[|x|]P == psi(P) (bls12-381.ts:599-601)
// in the lane-paired form - half the registers per lane instead of 587 spilled ones - and clears rejected rows.
__global__ void __launch_bounds__(128, NCG_DEC_G2_MINW) k_decode_g2_stage_a(const uint8_t* __restrict__ in, uint32_t* __restrict__ out,
                                                                            uint8_t* __restrict__ ok, uint8_t* __restrict__ inf, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t f = 0;
  ok[i] = g2_decode_lane<false>(in + (size_t)i * 96, out + (size_t)i * 48, &f) ? 1 : 0;
  inf[i] = f;
}
__global__ void __launch_bounds__(64, 2) k_decode_g2_stage_b(uint32_t* __restrict__ out, uint8_t* __restrict__ ok,
                                                             const uint8_t* __restrict__ inf, int n) {
  using F = FeBls2P;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;  // two lanes per point
  if (i >= n) return;
  if (!ok[i] || inf[i]) return;                                  // rejected already, or the (valid) infinity encoding
  uint32_t* row = out + (size_t)i * 48;
  const F x = FieldWire<F>::load(row), y = FieldWire<F>::load(row + 24);
  Jac<F> P{x, y, F::one()};
  Jac<F> xP = jac_neg(bls_mul_by_x(P));
  const Fe29x2P<1> psx = p2_const(ParamsBls29::PSI_X_C0, ParamsBls29::PSI_X_C1);
  const Fe29x2P<1> psy = p2_const(ParamsBls29::PSI_Y_C0, ParamsBls29::PSI_Y_C1);
  auto px = p2_conj(x) * psx;
  auto py = p2_conj(y) * psy;
  auto zz = f_sqr(xP.Z);
  auto dx = xP.X - px * zz;
  auto dy = xP.Y - py * zz * xP.Z;
  const bool same = !f_eqz(xP.Z) && f_eqz(dx) && f_eqz(dy);  // pair-uniform
  if (!same) {
    const int half = (threadIdx.x & 1) * 12;  // each lane clears its halves of x and y
#pragma unroll
    for (int k = 0; k < 12; k++) {
      row[half + k] = 0;
      row[24 + half + k] = 0;
    }
    if ((threadIdx.x & 1) == 0) ok[i] = 0;
  }
}

__global__ void __launch_bounds__(256) k_decode_ed(const uint8_t* __restrict__ in, int zip215,
                                                   uint32_t* __restrict__ out, uint8_t* __restrict__ ok, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ok[i] = ed_decode_lane(in + (size_t)i * 32, zip215 != 0, out + (size_t)i * 16) ? 1 : 0;
}

int decode_in_bytes(int curve) {
  switch (curve) {
    case CURVE_SECP256K1: return 33;
    case CURVE_ED25519: return 32;
    case CURVE_BLS12_381_G1: return 48;
    case CURVE_BLS12_381_G2: return 96;
    default: return 0;
  }
}

hipError_t decode_points_batch(int curve, const uint8_t* in, int flags, uint32_t* out, uint8_t* ok, uint8_t* inf, int n,
                               hipStream_t st) {
  if (n <= 0) return hipSuccess;
  dim3 grid((n + 255) / 256), block(256);
  switch (curve) {
    case CURVE_SECP256K1:
      hipLaunchKernelGGL(k_decode_secp, grid, block, 0, st, in, out, ok, n);
      (void)hipMemsetAsync(inf, 0, n, st);
      break;
    case CURVE_ED25519:
      hipLaunchKernelGGL(k_decode_ed, grid, block, 0, st, in, flags & 1, out, ok, n);
      (void)hipMemsetAsync(inf, 0, n, st);
      break;
    case CURVE_BLS12_381_G1:
      hipLaunchKernelGGL(k_decode_g1, grid, block, 0, st, in, out, ok, inf, n);
      break;
    case CURVE_BLS12_381_G2: {
#ifdef NCG_AB_BUILD  // the fused kernel (587 spilled registers) only exists in A/B builds
      static const int fused = knob("NCG_DEC_G2_FUSED", 0);
      if (fused) {
        hipLaunchKernelGGL(k_decode_g2, dim3((n + 127) / 128), dim3(128), 0, st, in, out, ok, inf, n);
        break;
      }
#endif
      hipLaunchKernelGGL(k_decode_g2_stage_a, dim3((n + 127) / 128), dim3(128), 0, st, in, out, ok, inf, n);
      hipLaunchKernelGGL(k_decode_g2_stage_b, dim3((unsigned)(((size_t)n * 2 + 63) / 64)), dim3(64), 0, st, out, ok, inf, n);
      break;
    }
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------- encoders
// Point.toBytes (compressed) of affine wire points: byte shuffles only, inputs are canonical.
//   secp256k1 pointToBytes (weierstrass.ts:541-564; ZERO is rejected -> ok = 0)
//   bls12-381 coder.encode (bls12-381.ts:400-410, sortBit :346-351, fp2.encode c1 || c0 :354-361)
//   ed25519   Point.toBytes (edwards.ts:620-628): y little-endian, bit 255 = x odd
NCG_DI void words_to_be(const uint32_t* __restrict__ w, int nw, uint8_t* __restrict__ out) {
  for (int i = 0; i < nw; i++) {
    const uint32_t v = w[nw - 1 - i];
    out[4 * i] = (uint8_t)(v >> 24);
    out[4 * i + 1] = (uint8_t)(v >> 16);
    out[4 * i + 2] = (uint8_t)(v >> 8);
    out[4 * i + 3] = (uint8_t)v;
  }
}
NCG_DI bool encode_lane(int curve, const uint32_t* __restrict__ in, uint8_t* __restrict__ out) {
  if (curve == CURVE_SECP256K1) {
    uint32_t any = 0;
    for (int i = 0; i < 16; i++) any |= in[i];
    if (any == 0) {
      for (int i = 0; i < 33; i++) out[i] = 0;
      return false;  // "bad point: ZERO"
    }
    out[0] = (in[8] & 1u) ? 3 : 2;
    words_to_be(in, 8, out + 1);
    return true;
  }
  if (curve == CURVE_ED25519) {
    for (int i = 0; i < 8; i++) {
      const uint32_t v = in[8 + i];
      out[4 * i] = (uint8_t)v;
      out[4 * i + 1] = (uint8_t)(v >> 8);
      out[4 * i + 2] = (uint8_t)(v >> 16);
      out[4 * i + 3] = (uint8_t)(v >> 24);
    }
    out[31] |= (uint8_t)((in[0] & 1u) << 7);
    return true;
  }
  const bool g2 = curve == CURVE_BLS12_381_G2;
  const int pw = g2 ? 48 : 24, nb = g2 ? 96 : 48;
  uint32_t any = 0;
  for (int i = 0; i < pw; i++) any |= in[i];
  if (any == 0) {  // infinity: compressed + infinity flags, zero payload
    out[0] = 0xC0;
    for (int i = 1; i < nb; i++) out[i] = 0;
    return true;
  }
  bool sort;
  if (g2) {
    words_to_be(in + 12, 12, out);  // x.c1
    words_to_be(in, 12, out + 48);  // x.c0
    uint32_t y0[12], y1[12], any1 = 0;
    for (int i = 0; i < 12; i++) {
      y0[i] = in[24 + i];
      y1[i] = in[36 + i];
      any1 |= y1[i];
    }
    sort = any1 ? words12_gt_half_p(y1) : words12_gt_half_p(y0);
  } else {
    words_to_be(in, 12, out);
    uint32_t y[12];
    for (int i = 0; i < 12; i++) y[i] = in[12 + i];
    sort = words12_gt_half_p(y);
  }
  out[0] |= 0x80 | (sort ? 0x20 : 0);
  return true;
}

__global__ void __launch_bounds__(256) k_encode(int curve, const uint32_t* __restrict__ in, int in_words,
                                                uint8_t* __restrict__ out, int out_bytes, uint8_t* __restrict__ ok,
                                                int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ok[i] = encode_lane(curve, in + (size_t)i * in_words, out + (size_t)i * out_bytes) ? 1 : 0;
}

hipError_t encode_points_batch(int curve, const uint32_t* in, uint8_t* out, uint8_t* ok, int n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  const int ob = decode_in_bytes(curve);
  if (ob == 0) return hipErrorInvalidValue;
  const int iw = curve == CURVE_BLS12_381_G2 ? 48 : curve == CURVE_BLS12_381_G1 ? 24 : 16;
  hipLaunchKernelGGL(k_encode, dim3((n + 255) / 256), dim3(256), 0, st, curve, in, iw, out, ob, ok, n);
  return hipGetLastError();
}
void encode_points_host(int curve, const uint32_t* in, uint8_t* out, uint8_t* ok, int n) {
  const int ob = decode_in_bytes(curve);
  const int iw = curve == CURVE_BLS12_381_G2 ? 48 : curve == CURVE_BLS12_381_G1 ? 24 : 16;
  for (int i = 0; i < n; i++) ok[i] = encode_lane(curve, in + (size_t)i * iw, out + (size_t)i * ob) ? 1 : 0;
}

// host-side execution of the lane functions (unit tests through hosttest.hip)
void decode_points_host(int curve, const uint8_t* in, int flags, uint32_t* out, uint8_t* ok, uint8_t* inf, int n) {
  for (int i = 0; i < n; i++) {
    inf[i] = 0;
    if (curve == CURVE_SECP256K1) ok[i] = secp_decode_lane(in + (size_t)i * 33, out + (size_t)i * 16);
    else if (curve == CURVE_ED25519) ok[i] = ed_decode_lane(in + (size_t)i * 32, (flags & 1) != 0, out + (size_t)i * 16);
    else if (curve == CURVE_BLS12_381_G1) ok[i] = g1_decode_lane(in + (size_t)i * 48, out + (size_t)i * 24, inf + i);
    else if (curve == CURVE_BLS12_381_G2) ok[i] = g2_decode_lane(in + (size_t)i * 96, out + (size_t)i * 48, inf + i);
    else ok[i] = 0;
  }
}

}  // namespace ncg
