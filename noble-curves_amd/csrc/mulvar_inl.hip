// secp256k1 batch multiply with the field multiply INLINED into the group-law routines (no call, no
// argument marshalling through v0..v17; the doubling loop stays rolled so the window body fits the
// instruction cache).  A/B alternative to the out-of-line build in mulvar.hip, selected by
// NCG_SECP_W=25x (x = waves/SIMD requested).  The curve twin gives the kernels their own names: both
// translation units ship their own code object.
#define NCG_MUL_INLINE 1
#include "mulvar.hpp"
#include "knobs.hpp"
#include "host_api.hpp"

#include <cstdlib>

namespace ncg {

struct CurveSecpI : CurveSecp {};

hipError_t mul_var_secp_inline(int minw, const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                               uint32_t* jac_tmp, hipStream_t st) {
#ifndef NCG_AB_BUILD
  if (minw == 13) return launch_mul_var_gtab<CurveSecpI, 4, 3, 16>(pts, scalars, out, out_inf, n, jac_tmp, st);  // the shipped kernel
  return hipErrorInvalidValue;
#else
  static const int k = knob("NCG_AFF_K", 16);
  switch (minw * 100 + k) {
    case 408: return launch_mul_var_gtab<CurveSecpI, 5, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
    case 308: return launch_mul_var_gtab<CurveSecpI, 5, 3>(pts, scalars, out, out_inf, n, jac_tmp, st);
    case 304: return launch_mul_var_gtab<CurveSecpI, 5, 3, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
    case 316: return launch_mul_var_gtab<CurveSecpI, 5, 3, 16>(pts, scalars, out, out_inf, n, jac_tmp, st);
    case 208: return launch_mul_var_gtab<CurveSecpI, 5, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);
    case 1308: return launch_mul_var_gtab<CurveSecpI, 4, 3>(pts, scalars, out, out_inf, n, jac_tmp, st);  // minw 13: W = 4
    case 1316: return launch_mul_var_gtab<CurveSecpI, 4, 3, 16>(pts, scalars, out, out_inf, n, jac_tmp, st);  // the default
    case 1332: return launch_mul_var_gtab<CurveSecpI, 4, 3, 32>(pts, scalars, out, out_inf, n, jac_tmp, st);
    case 1408: return launch_mul_var_gtab<CurveSecpI, 4, 4>(pts, scalars, out, out_inf, n, jac_tmp, st);
    default: return hipErrorInvalidValue;
  }
#endif
}

}  // namespace ncg
