// Fixed point sets: window-shifted copies for the shared-bucket MSM.
//
// The reference's interleavedMSMUnsafe (src/abstract/curve.ts:907-959) precomputes per-point tables ONCE for a
// point set and then takes only scalars.  The device analogue: for a resident set, level w holds 2^(c w) P for every
// point P (affine, the accumulate kernel's storage format).  Then window w of every scalar multiplies the level-w
// copy by its digit with weight ONE, so all windows add into the same 2^(c-1) buckets (msm.hpp `shared`): the bucket
// fold runs once instead of once per window and the Horner combine across windows (255 dependent doublings on the
// host, curve.ts:901-902) disappears - only the c doublings inside the single window remain.
// Cost: (nwin - 1) * c doublings per point, once, and nwin copies of the set in HBM (16 x 112 B x 2^20 = 1.9 GB for
// G1 - sized for 288 GB).  Endomorphism images (endo.hpp) commute with doubling, so verified sets shift their
// expanded image array the same way.
#include "host_api.hpp"
#include "mulvar.hpp"
#include "msm.hpp"

namespace ncg {

// stored affine (x, y) -> c doublings -> Jacobian [m][3][FW] (Z = 0 for the identity)
template <class C>
__global__ void __launch_bounds__(256) k_shift_dbl(const uint32_t* __restrict__ in, uint32_t* __restrict__ jac, int m, int c) {
  using F = typename C::F;
  constexpr int FW = FieldIO<F>::WORDS;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> LaneShift<C>::value;
  if (i >= m) return;
  const uint32_t* p = in + (size_t)i * 2 * FW;
  Affine<F> a{FieldIO<F>::load(p), FieldIO<F>::load(p + FW)};
  Jac<F> J = jac_from_affine(a);
  for (int k = 0; k < c; k++) J = jac_dbl(J);
  uint32_t* o = jac + (size_t)i * 3 * FW;
  FieldIO<F>::store(o, J.X);
  FieldIO<F>::store(o + FW, J.Y);
  FieldIO<F>::store(o + 2 * FW, J.Z);
}

template <class C, int K>
static hipError_t shift_level_t(int curve, const uint32_t* d_prev, int m, int c, uint32_t* d_jac, uint32_t* d_wire, uint8_t* d_inf,
                                uint32_t* d_out, hipStream_t st) {
  constexpr int LS = LaneShift<C>::value;
  hipLaunchKernelGGL(k_shift_dbl<C>, dim3((unsigned)((((size_t)m << LS) + 255) / 256)), dim3(256), 0, st, d_prev, d_jac, m, c);
  const int threads = ((m + K - 1) / K) << LS;
  hipLaunchKernelGGL((k_jac_batch_affine<C, K>), dim3((threads + 255) / 256), dim3(256), 0, st, d_jac, d_wire, d_inf, m);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return msm_points_to_stored(curve, d_wire, m, d_out, st);
}

size_t msm_shift_tmp_bytes(int curve, int m) {
  const size_t sw = msm_stored_words_per_point(curve);   // 2 FW
  const size_t wire = curve == CURVE_BLS12_381_G2 ? 48 : curve == CURVE_BLS12_381_G1 ? 24 : 16;
  return ((size_t)m * (sw / 2 * 3) * 4 + 255) / 256 * 256 + ((size_t)m * wire * 4 + 255) / 256 * 256 + (size_t)m + 256;
}

// levels[1 .. nlev) from levels[0] (m stored affine points each, level stride m * stored words); tmp: msm_shift_tmp_bytes
hipError_t msm_shift_levels(int curve, uint32_t* d_levels, int m, int nlev, int c, void* d_tmp, hipStream_t st) {
  const size_t sw = msm_stored_words_per_point(curve);
  const size_t wire = curve == CURVE_BLS12_381_G2 ? 48 : curve == CURVE_BLS12_381_G1 ? 24 : 16;
  char* t = (char*)d_tmp;
  uint32_t* d_jac = (uint32_t*)t;
  t += ((size_t)m * (sw / 2 * 3) * 4 + 255) / 256 * 256;
  uint32_t* d_wire = (uint32_t*)t;
  t += ((size_t)m * wire * 4 + 255) / 256 * 256;
  uint8_t* d_inf = (uint8_t*)t;
  for (int w = 1; w < nlev; w++) {
    const uint32_t* prev = d_levels + (size_t)(w - 1) * m * sw;
    uint32_t* out = d_levels + (size_t)w * m * sw;
    hipError_t e;
    switch (curve) {
      case CURVE_SECP256K1: e = shift_level_t<CurveSecp, 16>(curve, prev, m, c, d_jac, d_wire, d_inf, out, st); break;
      case CURVE_BLS12_381_G1: e = shift_level_t<CurveG1, 8>(curve, prev, m, c, d_jac, d_wire, d_inf, out, st); break;
      case CURVE_BLS12_381_G2: e = shift_level_t<CurveG2P, 4>(curve, prev, m, c, d_jac, d_wire, d_inf, out, st); break;
      default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace ncg
