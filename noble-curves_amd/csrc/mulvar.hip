// Launchers for the batch variable-base multiply kernels (see mulvar.hpp).
#include <algorithm>
#include "mulvar.hpp"
#include "knobs.hpp"
#include "host_api.hpp"

#include <cstdlib>

namespace ncg {

// `jac_tmp` (n * 3 * FW words, device) enables the two-kernel path: ladder -> Jacobian, then
// one batched inversion per K points.  Without it every lane inverts its own Z.
template <class C, int W, int MINW = 1, int K = 8>
static hipError_t launch_mul_var(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf,
                                 int n, uint32_t* jac_tmp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  using Cfg = MulVarCfg<C, W>;
  constexpr int LS = LaneShift<C>::value;  // lanes per item = 1 << LS
  const unsigned blocks = (unsigned)((((size_t)n << LS) + 63) / 64);
  size_t lds = (size_t)Cfg::LDS_WORDS * 4;
  if (jac_tmp) {
    auto kern = k_mul_var<C, W, MINW, true>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, st, pts, scalars, jac_tmp, out_inf, n);
    int threads = ((n + K - 1) / K) << LS;
    hipLaunchKernelGGL((k_jac_batch_affine<C, K>), dim3((threads + 255) / 256), dim3(256), 0, st, jac_tmp, out, out_inf,
                       n);
    return hipGetLastError();
  }
  auto kern = k_mul_var<C, W, MINW, false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, st, pts, scalars, out, out_inf, n);
  return hipGetLastError();
}

// ---- pairwise addition out[i] = A[i] + B[i] (or A[i] - B[i]) ---------------------------------------
// Point.add / subtract of the reference for a batch of pairs (src/abstract/weierstrass.ts:834-891 incl.
// the P = Q, P = -Q and ZERO cases; src/abstract/edwards.ts:526-545), and the combining step of
// Point.mulAddUnsafe (weierstrass.ts:937-944: a*P + b*Q = two batch multiplies + this).
template <class C>
__global__ void __launch_bounds__(256) k_pair_add(const uint32_t* __restrict__ a_wire, const uint32_t* __restrict__ b_wire,
                                                  int subtract, uint32_t* __restrict__ jac_out, int n) {
  using F = typename C::F;
  constexpr int FW = FieldIO<F>::WORDS, WW = FieldWire<F>::WORDS;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> LaneShift<C>::value;
  if (i >= n) return;
  Affine<F> A = load_affine_wire<F>(a_wire + (size_t)i * 2 * WW);
  Affine<F> B = load_affine_wire<F>(b_wire + (size_t)i * 2 * WW);
  if (subtract) B.y = f_neg(B.y);
  Jac<F> R = jac_madd(jac_from_affine(A), B);
  if (R.is_inf()) R = Jac<F>::inf();
  uint32_t* o = jac_out + (size_t)i * 3 * FW;
  FieldIO<F>::store(o, R.X);
  FieldIO<F>::store(o + FW, R.Y);
  FieldIO<F>::store(o + 2 * FW, R.Z);
}
__global__ void __launch_bounds__(256) k_pair_add_ed(const uint32_t* __restrict__ a_wire, const uint32_t* __restrict__ b_wire,
                                                     int subtract, uint32_t* __restrict__ proj_out, int n) {
  using F = FEd;
  constexpr int FW = FieldIO<F>::WORDS;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F ax = FieldWire<F>::load(a_wire + (size_t)i * 16), ay = FieldWire<F>::load(a_wire + (size_t)i * 16 + 8);
  F bx = FieldWire<F>::load(b_wire + (size_t)i * 16), by = FieldWire<F>::load(b_wire + (size_t)i * 16 + 8);
  EdExt<F> A{ax, ay, F::one(), ax * ay};
  EdExt<F> R = ed_madd_niels(A, ed_affine_to_niels(bx, by, EdConsts::d2()), subtract != 0);
  uint32_t* o = proj_out + (size_t)i * 3 * FW;
  FieldIO<F>::store(o, R.X);
  FieldIO<F>::store(o + FW, R.Y);
  FieldIO<F>::store(o + 2 * FW, R.Z);
}

template <class C, int K>
static hipError_t pair_add_t(const uint32_t* a, const uint32_t* b, int subtract, uint32_t* out, uint8_t* out_inf, int n,
                             uint32_t* jac_tmp, hipStream_t st) {
  constexpr int LS = LaneShift<C>::value;
  hipLaunchKernelGGL(k_pair_add<C>, dim3((unsigned)((((size_t)n << LS) + 255) / 256)), dim3(256), 0, st, a, b, subtract,
                     jac_tmp, n);
  int threads = ((n + K - 1) / K) << LS;
  hipLaunchKernelGGL((k_jac_batch_affine<C, K>), dim3((threads + 255) / 256), dim3(256), 0, st, jac_tmp, out, out_inf, n);
  return hipGetLastError();
}

// jac_tmp: mul_var_tmp_bytes(curve, n) bytes of device scratch
hipError_t pair_add_batch(int curve, const uint32_t* a, const uint32_t* b, int subtract, uint32_t* out, uint8_t* out_inf,
                          int n, uint32_t* jac_tmp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  switch (curve) {
    case CURVE_SECP256K1: return pair_add_t<CurveSecp, 8>(a, b, subtract, out, out_inf, n, jac_tmp, st);
    case CURVE_BLS12_381_G1: return pair_add_t<CurveG1, 8>(a, b, subtract, out, out_inf, n, jac_tmp, st);
    case CURVE_BLS12_381_G2: return pair_add_t<CurveG2P, 4>(a, b, subtract, out, out_inf, n, jac_tmp, st);
    case CURVE_ED25519:
      hipLaunchKernelGGL(k_pair_add_ed, dim3((n + 255) / 256), dim3(256), 0, st, a, b, subtract, jac_tmp, n);
      return ed25519_proj_to_affine(jac_tmp, out, out_inf, n, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t normalize_batch(int curve, const uint32_t* proj_wire, uint32_t* out_wire, uint8_t* out_inf, int n,
                           hipStream_t st) {
  if (n <= 0) return hipSuccess;
  switch (curve) {
    case CURVE_SECP256K1:
      hipLaunchKernelGGL((k_proj_batch_affine<FpSecp, 8>), dim3(((n + 7) / 8 + 255) / 256), dim3(256), 0, st, proj_wire,
                         out_wire, out_inf, n);
      break;
    case CURVE_ED25519:
      hipLaunchKernelGGL((k_proj_batch_affine<FpEd, 8>), dim3(((n + 7) / 8 + 255) / 256), dim3(256), 0, st, proj_wire,
                         out_wire, out_inf, n);
      break;
    case CURVE_BLS12_381_G1:
      hipLaunchKernelGGL((k_proj_batch_affine<FeBls, 8>), dim3(((n + 7) / 8 + 255) / 256), dim3(256), 0, st, proj_wire,
                         out_wire, out_inf, n);
      break;
    case CURVE_BLS12_381_G2:
      hipLaunchKernelGGL((k_proj_batch_affine<FeBls2, 4>), dim3(((n + 3) / 4 + 255) / 256), dim3(256), 0, st, proj_wire,
                         out_wire, out_inf, n);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

size_t mul_var_tmp_bytes(int curve, int n) {
  switch (curve) {
    // Jacobian scratch + the per-item window table of the widest variant (k_mul_var_gtab)
    case CURVE_SECP256K1: return pad64(n) * (3 * FieldIO<CurveSecp::F>::WORDS + gtab_words_per_item<CurveSecp, 5>()) * 4;
    case CURVE_BLS12_381_G1: return pad64(n) * (3 * FieldIO<CurveG1::F>::WORDS + gtab_words_per_item<CurveG1, 5>()) * 4;
    case CURVE_BLS12_381_G2:  // the verified-set ladder (mulvar_endo.hip) keeps a second table per lane
      return std::max(pad64(n) * (3 * FieldIO<CurveG2::F>::WORDS + 2 * gtab_words_per_item<CurveG2P, 4>()) * 4,
                      mul_var_g2_subgroup_tmp_bytes(n));
    case CURVE_ED25519:  // (X, Y, Z) + per-item window table of the multiply; the two tables per item of the verification
      return std::max(ed25519_tmp_words(n), ed25519_verify_tmp_words(n)) * 4;
    default: return 0;
  }
}

hipError_t mul_var_batch(int curve, const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf,
                         int n, uint32_t* jac_tmp, hipStream_t st) {
  switch (curve) {
    // Window width / occupancy / table placement, chosen by A/B runs on MI355X (DESIGN.md section 5;
    // NCG_SECP_W / NCG_G1_W / NCG_G2_W select the alternatives: 1WM = table in device memory with
    // W-bit windows and M waves/SIMD requested, WM = table in LDS).
    case CURVE_SECP256K1: {
#ifdef NCG_AB_BUILD  // the measured alternatives (tools/ab_secp.sh builds with -DNCG_AB_BUILD): not in the shipped library
      static const int w = knob("NCG_SECP_W", 243);
      if (jac_tmp && w == 154) return launch_mul_var_gtab<CurveSecp, 5, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w == 153) return launch_mul_var_gtab<CurveSecp, 5, 3>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w == 152) return launch_mul_var_gtab<CurveSecp, 5, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w >= 252 && w <= 254) return mul_var_secp_inline(w - 250, pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w >= 243 && w <= 244) return mul_var_secp_inline(w - 230, pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w == 144) return launch_mul_var_gtab<CurveSecp, 4, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w == 133) return launch_mul_var_gtab<CurveSecp, 3, 3>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (w == 42) return launch_mul_var<CurveSecp, 4, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);
#else
      if (jac_tmp) return mul_var_secp_inline(13, pts, scalars, out, out_inf, n, jac_tmp, st);  // W = 4, 3 waves/SIMD, multiply inlined
#endif
      return launch_mul_var<CurveSecp, 3, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);  // LDS table (also without scratch)
    }
    case CURVE_ED25519: return ed25519_mul_var_batch(pts, scalars, out, out_inf, n, jac_tmp, st);
    case CURVE_BLS12_381_G1: {
#ifdef NCG_AB_BUILD
      static const int w = knob("NCG_G1_W", 142);
      if (jac_tmp && w == 141) return launch_mul_var_gtab<CurveG1, 4, 1>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w == 152) return launch_mul_var_gtab<CurveG1, 5, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w == 151) return launch_mul_var_gtab<CurveG1, 5, 1>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w == 132) return launch_mul_var_gtab<CurveG1, 3, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w == 142) return launch_mul_var_gtab<CurveG1, 4, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);
#else
      if (jac_tmp) return launch_mul_var_gtab<CurveG1, 4, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);
#endif
      return launch_mul_var<CurveG1, 3, 1, 8>(pts, scalars, out, out_inf, n, jac_tmp, st);
    }
    case CURVE_BLS12_381_G2: {
#ifdef NCG_AB_BUILD
      static const int w = knob("NCG_G2_W", 142);
      if (w == 0) return launch_mul_var<CurveG2, 3, 1, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);  // unpaired
      if (jac_tmp && w == 142) return launch_mul_var_gtab<CurveG2P, 4, 2, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
#else
      if (jac_tmp) return launch_mul_var_gtab<CurveG2P, 4, 2, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
#endif
      return launch_mul_var<CurveG2P, 2, 2, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
    }
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ncg
