// Launchers for the batch variable-base multiply kernels (see mulvar.hpp).
#include "mulvar.hpp"
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

// Table in device memory (k_mul_var_gtab): the table lives behind the Jacobian scratch in `jac_tmp`
// (mul_var_tmp_bytes accounts for both).
template <class C, int W>
static size_t gtab_words_per_item() { return (size_t)MulVarCfg<C, W>::TS * 3 * MulVarCfg<C, W>::TW; }
static size_t pad64(size_t n) { return (n + 63) / 64 * 64; }

template <class C, int W, int MINW, int K = 8>
static hipError_t launch_mul_var_gtab(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf,
                                      int n, uint32_t* jac_tmp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  using Cfg = MulVarCfg<C, W>;
  constexpr int LS = LaneShift<C>::value;
  uint32_t* gtab = jac_tmp + pad64(n) * 3 * Cfg::FW;
  const unsigned blocks = (unsigned)((((size_t)n << LS) + 63) / 64);
  hipLaunchKernelGGL((k_mul_var_gtab<C, W, MINW, true>), dim3(blocks), dim3(64), 0, st, pts, scalars, jac_tmp, out_inf,
                     gtab, n);
  int threads = ((n + K - 1) / K) << LS;
  hipLaunchKernelGGL((k_jac_batch_affine<C, K>), dim3((threads + 255) / 256), dim3(256), 0, st, jac_tmp, out, out_inf, n);
  return hipGetLastError();
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
    case CURVE_BLS12_381_G1: return pad64(n) * (3 * FieldIO<CurveG1::F>::WORDS + gtab_words_per_item<CurveG1, 4>()) * 4;
    case CURVE_BLS12_381_G2: return pad64(n) * (3 * FieldIO<CurveG2::F>::WORDS + 2 * gtab_words_per_item<CurveG2P, 4>()) * 4;
    case CURVE_ED25519: return ed25519_tmp_words(n) * 4;  // (X, Y, Z) + per-item window tables
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
      static const int w = [] { const char* e = std::getenv("NCG_SECP_W"); return e ? std::atoi(e) : 154; }();
      if (jac_tmp && w == 154) return launch_mul_var_gtab<CurveSecp, 5, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w == 144) return launch_mul_var_gtab<CurveSecp, 4, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w == 133) return launch_mul_var_gtab<CurveSecp, 3, 3>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (w == 42) return launch_mul_var<CurveSecp, 4, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);
      return launch_mul_var<CurveSecp, 3, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);  // LDS table (also without scratch)
    }
    case CURVE_ED25519: return ed25519_mul_var_batch(pts, scalars, out, out_inf, n, jac_tmp, st);
    case CURVE_BLS12_381_G1: {
      static const int w = [] { const char* e = std::getenv("NCG_G1_W"); return e ? std::atoi(e) : 141; }();
      if (jac_tmp && w == 141) return launch_mul_var_gtab<CurveG1, 4, 1>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (jac_tmp && w == 132) return launch_mul_var_gtab<CurveG1, 3, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);
      return launch_mul_var<CurveG1, 3, 1, 8>(pts, scalars, out, out_inf, n, jac_tmp, st);
    }
    case CURVE_BLS12_381_G2: {
      static const int w = [] { const char* e = std::getenv("NCG_G2_W"); return e ? std::atoi(e) : 142; }();
      if (jac_tmp && w == 142) return launch_mul_var_gtab<CurveG2P, 4, 2, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
      if (w == 0) return launch_mul_var<CurveG2, 3, 1, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);  // unpaired
      return launch_mul_var<CurveG2P, 2, 2, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
    }
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ncg
