// Endomorphism mode of the bls12-381 MSM (endo.hpp): the window plan, the digit kernel (scalar split +
// signed windows), the kernel that expands a verified point set into its endomorphism images (stored in the
// accumulate kernel's input format, so the per-call wire -> Montgomery conversion disappears too) and the
// one-off subgroup verification of a raw point set.  Everything downstream of the digits (sort, bucket
// accumulate, fold, finish) is the unchanged pipeline of msm.hip running on endo * n entries.
#include <algorithm>
#include "knobs.hpp"
#include <cstdlib>

#include "bls_lanes.hpp"
#include "endo.hpp"
#include "host_api.hpp"
#include "msm.hpp"
#include "msm_plan.hpp"

namespace ncg {

// ------------------------------------------------------------------ digits
// digits[w * n + e * n_src + i] = (((sub_e(k_i) + H') >> (c w)) & (2^c - 1)) - 2^(c-1)
template <int E>
__global__ void __launch_bounds__(256) k_msm_digits_endo(const uint32_t* __restrict__ scalars, int16_t* __restrict__ digits,
                                                         MsmPlan pl, uint32_t* __restrict__ bad_index) {
  __shared__ uint32_t sh[256 * 8];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pl.n_src) return;
  uint32_t k[8];
  {
    const uint4* kp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
    const uint4 lo = kp[0], hi = kp[1];
    k[0] = lo.x; k[1] = lo.y; k[2] = lo.z; k[3] = lo.w;
    k[4] = hi.x; k[5] = hi.y; k[6] = hi.z; k[7] = hi.w;
  }
  {
    uint32_t bw = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) (void)__builtin_subc(k[j], pl.order[j], bw, &bw);
    if (bw == 0) atomicMin(bad_index, (uint32_t)i);  // scalar >= order (validateMSMScalars, curve.ts:398-404)
  }
  uint32_t sub[E][6];
  if constexpr (E == 2) bls_endo_split2(sub, k);
  else bls_endo_split4(sub, k);
  uint32_t* my = sh + threadIdx.x * 8;
  const uint32_t mask = (1u << pl.c) - 1u;
  const int half = 1 << (pl.c - 1);
#pragma unroll
  for (int e = 0; e < E; e++) {
    uint32_t cy = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) my[j] = __builtin_addc(sub[e][j], pl.hconst[j], cy, &cy);
    my[6] = 0;
    my[7] = 0;
    for (int w = 0; w < pl.nwin; w++) {
      const int bp = (w + pl.w0) * pl.c;
      const int limb = bp >> 5, sft = bp & 31;
      const uint64_t two = ((uint64_t)my[limb + 1] << 32) | my[limb];
      const uint32_t v = (uint32_t)(two >> sft) & mask;
      digits[(size_t)w * pl.n + (size_t)e * pl.n_src + i] = (int16_t)((int)v - half);
    }
  }
}

hipError_t msm_endo_digits(const MsmPlan& pl, const uint32_t* d_scalars, int16_t* digits, uint32_t* bad, hipStream_t st) {
  const dim3 grid((pl.n_src + 255) / 256), block(256);
  if (pl.endo == 2) hipLaunchKernelGGL(k_msm_digits_endo<2>, grid, block, 0, st, d_scalars, digits, pl, bad);
  else if (pl.endo == 4) hipLaunchKernelGGL(k_msm_digits_endo<4>, grid, block, 0, st, d_scalars, digits, pl, bad);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ------------------------------------------------------------------ plan
int msm_endo_factor(int curve) { return curve == CURVE_BLS12_381_G1 ? 2 : curve == CURVE_BLS12_381_G2 ? 4 : 0; }

static int ilog2u(unsigned x) {
  int r = 0;
  while (x >>= 1) r++;
  return r;
}

int msm_make_plan_endo(int curve, int n_src, int c_override, MsmPlan* pl) {
  const int E = msm_endo_factor(curve);
  if (E == 0 || n_src <= 0 || (long)n_src * E > 0x7fffffffL) return -1;
  const int bits = E == 2 ? 128 : 64;  // |sub-scalar| <= 2^(bits-1) * 0.68 (endo.hpp)
  const int n = n_src * E;
  int c = c_override;
  if (c <= 0) {
    c = knob("NCG_MSM_C_ENDO", 0);
  }
  if (c <= 0) {
    // as in msm_plan.hpp: the width also decides how full the TOP window is - 128-bit sub-scalars in 14-bit windows
    // leave it 2 bits, i.e. a handful of buckets holding every entry of the window (very long fix-up runs; measured:
    // G1 2^17 verified 1.67 ms at c = 14 against 1.06 ms for the generic path).  Take the width closest to
    // log2(entries) - 4 (G2: - 3) whose top window is at least 80 % full.
    const int c0 = std::max(3, std::min(16, ilog2u((unsigned)n) - (curve == CURVE_BLS12_381_G2 ? 3 : 4)));
    int best = c0, best_d = 99;
    for (int cc = std::max(3, c0 - 3); cc <= std::min(16, c0 + 3); cc++) {
      const int nw = (bits + cc - 1) / cc, top = bits - (nw - 1) * cc;
      if (top * 5 < cc * 4) continue;
      const int d = cc > c0 ? cc - c0 : c0 - cc;
      if (d < best_d) {
        best_d = d;
        best = cc;
      }
    }
    c = best;
  }
  // c >= 3 keeps |sub| + H' below 2^(c nwin) even when c nwin == bits: H' < 2^(c nwin - 1) (1 + 1/(2^c - 1))
  c = std::max(3, std::min(16, c));
  *pl = MsmPlan();
  pl->n = n;
  pl->n_src = n_src;
  pl->endo = E;
  pl->ls = curve == CURVE_BLS12_381_G2 ? 1 : 0;
  pl->accum_waves = curve == CURVE_BLS12_381_G2 ? NCG_G2_ACCUM_WAVES : 2;
  pl->c = c;
  pl->nb = 1 << (c - 1);
  pl->nwin = (bits + c - 1) / c;
  for (int i = 0; i < 10; i++) pl->hconst[i] = 0;
  for (int w = 0; w < pl->nwin; w++) {
    const int bit = c * w + c - 1;
    pl->hconst[bit >> 5] |= 1u << (bit & 31);
  }
  static const uint32_t BLS_R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  for (int i = 0; i < 8; i++) pl->order[i] = BLS_R[i];
  static const int q_blocks = std::max(64, knob("NCG_MSM_QBLOCKS", 512));
  int Q = std::max(1, q_blocks / pl->nwin);
  Q = std::min(Q, std::max(1, n / 4096));
  pl->Q = Q;
  pl->chunk = (n + Q - 1) / Q;
  msm_plan_sort_locality(*pl);
  return 0;
}

// ------------------------------------------------------------------ images
template <class T>
NCG_DI void endo_put(uint32_t* p, const T& v) {
  FieldIO<T>::store(p, v);
}
// out[(e * n + i)] = z^e P_i in the accumulate kernel's input format (x then y, Montgomery limbs).
// Infinity (wire all-zero) keeps all-zero images.
__global__ void __launch_bounds__(256) k_points_endo_g1(const uint32_t* __restrict__ pts, uint32_t* __restrict__ out, int n) {
  using F = FeBls;
  constexpr int FW = FieldIO<F>::WORDS, AW = 2 * FW;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine<F> p = load_affine_wire<F>(pts + (size_t)i * 24);
  uint32_t* o0 = out + (size_t)i * AW;
  uint32_t* o1 = out + ((size_t)n + i) * AW;
  FieldIO<F>::store(o0, p.x);
  FieldIO<F>::store(o0 + FW, p.y);
  if (p.is_inf()) {
#pragma unroll
    for (int j = 0; j < AW; j++) o1[j] = 0;
    return;
  }
  const Fe29<1> beta = fe29_const(ParamsBls29::G1_BETA);
  const auto bx = p.x * beta;       // z^2 P = -phi(P) = (beta x, -y)   (bls12-381.ts:567-577)
  const auto ny = f_neg(p.y);
  endo_put(o1, bx);
  endo_put(o1 + FW, ny);
}

__global__ void __launch_bounds__(128) k_points_endo_g2(const uint32_t* __restrict__ pts, uint32_t* __restrict__ out, int n) {
  using F = FeBls2;
  constexpr int FW = FieldIO<F>::WORDS, AW = 2 * FW;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine<F> p = load_affine_wire<F>(pts + (size_t)i * 48);
  uint32_t* o[4];
#pragma unroll
  for (int e = 0; e < 4; e++) o[e] = out + ((size_t)e * n + i) * AW;
  FieldIO<F>::store(o[0], p.x);
  FieldIO<F>::store(o[0] + FW, p.y);
  if (p.is_inf()) {
    for (int e = 1; e < 4; e++)
      for (int j = 0; j < AW; j++) o[e][j] = 0;
    return;
  }
  // psi(x, y) = (conj(x) PSI_X, conj(y) PSI_Y), psi^2(x, y) = (x PSI2_X, -y)  (tower.ts:242-256)
  const Fe29x2<1> psx{fe29_const(ParamsBls29::PSI_X_C0), fe29_const(ParamsBls29::PSI_X_C1)};
  const Fe29x2<1> psy{fe29_const(ParamsBls29::PSI_Y_C0), fe29_const(ParamsBls29::PSI_Y_C1)};
  const Fe29<1> k2 = fe29_const(BlsH2c::PSI2_X);
  const Fe29x2<2> x{p.x.c0 * Fe29<1>::one(), p.x.c1 * Fe29<1>::one()};  // tighten the storage bound
  const Fe29x2<2> y{p.y.c0 * Fe29<1>::one(), p.y.c1 * Fe29<1>::one()};
  const Fe29x2<2> cx{x.c0, f_neg(x.c1)}, cy{y.c0, f_neg(y.c1)};
  const auto a = cx * psx;  // psi(P).x
  const auto b = cy * psy;  // psi(P).y
  const auto nb = f_neg(b);
  // z P = -psi(P)
  endo_put(o[1], a);
  endo_put(o[1] + FW, nb);
  // z^2 P = psi^2(P)
  const Fe29x2<2> x2{x.c0 * k2, x.c1 * k2};
  const auto ny = f_neg(y);
  endo_put(o[2], x2);
  endo_put(o[2] + FW, ny);
  // z^3 P = -psi^3(P) = -psi(psi^2(P)) = (conj(x PSI2_X) PSI_X, conj(y) PSI_Y)
  const Fe29x2<2> a2{a.c0 * k2, a.c1 * k2};
  endo_put(o[3], a2);
  endo_put(o[3] + FW, b);
}

size_t msm_endo_words_per_point(int curve) {
  return curve == CURVE_BLS12_381_G1 ? 2 * FieldIO<FeBls>::WORDS : curve == CURVE_BLS12_381_G2 ? 2 * FieldIO<FeBls2>::WORDS : 0;
}

hipError_t msm_endo_expand(int curve, const uint32_t* d_pts_wire, int n, uint32_t* d_out, hipStream_t st) {
  if (curve == CURVE_BLS12_381_G1) hipLaunchKernelGGL(k_points_endo_g1, dim3((n + 255) / 256), dim3(256), 0, st, d_pts_wire, d_out, n);
  else if (curve == CURVE_BLS12_381_G2) hipLaunchKernelGGL(k_points_endo_g2, dim3((n + 127) / 128), dim3(128), 0, st, d_pts_wire, d_out, n);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ------------------------------------------------------------------ one-off subgroup verification
// P is in the prime-order subgroup iff z^2 P == -phi(P) on G1 (bls12-381.ts:567-577) / z P == -psi(P) on G2
// (:599-601): `mult` holds [z^2] P_i resp. [z] P_i computed by the generic batch multiply (affine wire +
// infinity flags), `images` the expanded set; *bad = smallest index that fails.
template <class F, int WW>
__global__ void __launch_bounds__(128) k_endo_verify(const uint32_t* __restrict__ mult, const uint8_t* __restrict__ mult_inf,
                                                     const uint32_t* __restrict__ images, int n, uint32_t* __restrict__ bad) {
  constexpr int FW = FieldIO<F>::WORDS, AW = 2 * FW;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* im = images + ((size_t)n + i) * AW;  // image 1
  uint32_t any = 0;
  for (int j = 0; j < AW; j++) any |= im[j];
  bool ok;
  if (any == 0) {
    ok = mult_inf[i] != 0;  // P = O
  } else {
    const F ix = FieldIO<F>::load(im), iy = FieldIO<F>::load(im + FW);
    uint32_t w[2 * WW];
    FieldWire<F>::store(w, ix);
    FieldWire<F>::store(w + WW, iy);
    uint32_t diff = 0;
    for (int j = 0; j < 2 * WW; j++) diff |= w[j] ^ mult[(size_t)i * 2 * WW + j];
    ok = diff == 0 && mult_inf[i] == 0;
  }
  if (!ok) atomicMin(bad, (uint32_t)i);
}

hipError_t msm_endo_verify(int curve, const uint32_t* d_mult, const uint8_t* d_mult_inf, const uint32_t* d_images, int n,
                           uint32_t* d_bad, hipStream_t st) {
  const dim3 grid((n + 127) / 128), block(128);
  if (curve == CURVE_BLS12_381_G1) hipLaunchKernelGGL((k_endo_verify<FeBls, 12>), grid, block, 0, st, d_mult, d_mult_inf, d_images, n, d_bad);
  else if (curve == CURVE_BLS12_381_G2) hipLaunchKernelGGL((k_endo_verify<FeBls2, 24>), grid, block, 0, st, d_mult, d_mult_inf, d_images, n, d_bad);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// the multiplier of the verification: z^2 (G1) or z (G2) as a 32-byte little-endian scalar
void msm_endo_verify_scalar(int curve, uint32_t (&k)[8]) {
  for (int i = 0; i < 8; i++) k[i] = 0;
  if (curve == CURVE_BLS12_381_G1)
    for (int i = 0; i < 4; i++) k[i] = BlsEndo::X2[i];
  else
    for (int i = 0; i < 2; i++) k[i] = BlsEndo::Z[i];
}

}  // namespace ncg
