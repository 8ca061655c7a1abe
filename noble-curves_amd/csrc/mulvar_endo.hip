// bls12-381 G1 batch multiply for points KNOWN to lie in the prime-order subgroup (resident sets that passed or
// were decoded with the reference's isTorsionFree): the GLV ladder of mulvar.hpp on CurveG1E (curves.hpp) -
// k = k1 + k2 z^2 with 128-bit halves, 33 windows x (4 doublings + 2 additions) instead of 65 x (4 + 1).
// Same table build, same small-order fallback and the same outputs as the generic kernel.
#include "mulvar.hpp"
#include "host_api.hpp"

namespace ncg {

hipError_t mul_var_batch_g1_subgroup(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                                     uint32_t* jac_tmp, hipStream_t st) {
  return launch_mul_var_gtab<CurveG1E, 4, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);
}

}  // namespace ncg
