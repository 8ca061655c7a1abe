// Host finish of the MSM: Horner over the grouped window sums, then canonical affine output
// (src/abstract/curve.ts:901-902 + weierstrass.ts:951-969).  fin: [ngroups][nwin] grouped sums V_j
// (k_msm_tail): window sum W_w = sum_j 2^(g j) V_j, result = sum_w 2^(c w) W_w - c doublings per window in total,
// one addition per group.  bls12-381 (G1, G2) runs the 64-bit-limb Jacobian form of bls_host64.hpp; the other
// curves (and NCG_MSM_HOST64=0, the A/B switch) run the device templates compiled for the host.
#pragma once
#include <cstdlib>
#include "knobs.hpp"

#include "bls_host64.hpp"
#include "msm.hpp"

namespace ncg {

constexpr int MSM_GROUP = 3;
inline int msm_ngroups(int c) { return c >= 2 ? (c - 1 + MSM_GROUP - 1) / MSM_GROUP : 1; }

template <class C>
inline void msm_host_finish(const uint32_t* fin, int c, int nwin, uint32_t* out_affine, uint8_t* out_inf) {
  using G = MsmGroup<C>;
  constexpr int XW = G::ACC_WORDS;
  const int ng = msm_ngroups(c);
  auto at = [&](int j, int w) { return G::acc_load(fin + ((size_t)j * nwin + w) * XW); };
  typename G::Acc acc = G::identity();
  for (int w = nwin - 1; w >= 0; w--) {
    for (int j = ng - 1; j >= 0; j--) {
      const int shift = j == ng - 1 ? c - MSM_GROUP * j : MSM_GROUP;
      for (int d = 0; d < shift; d++) acc = G::dbl(acc);
      acc = G::add(acc, at(j, w));
    }
  }
  G::to_affine_wire(acc, out_affine, out_inf);
}

inline bool msm_host64_enabled() {
  return knob("NCG_MSM_HOST64", 1) != 0;
}
// helper threads of the bls12-381 finish (bls_host64.hpp FinishPool): on unless NCG_NO_FINISH_THREADS is set
inline bool msm_finish_threads_enabled() {
  static const bool off = knob_set("NCG_NO_FINISH_THREADS");
  return !off;
}

template <class C>
inline void msm_host_finish_any(const uint32_t* fin, int c, int nwin, uint32_t* out_affine, uint8_t* out_inf) {
  msm_host_finish<C>(fin, c, nwin, out_affine, out_inf);
}
template <>
inline void msm_host_finish_any<CurveG1>(const uint32_t* fin, int c, int nwin, uint32_t* out_affine, uint8_t* out_inf) {
  if (msm_host64_enabled() && !msm_finish_threads_enabled()) h64::msm_finish_serial<h64::Fp>(fin, c, nwin, MSM_GROUP, msm_ngroups(c), 14, 12, out_affine, out_inf);
  else if (msm_host64_enabled()) h64::msm_finish<h64::Fp>(fin, c, nwin, MSM_GROUP, msm_ngroups(c), 14, 12, out_affine, out_inf);
  else msm_host_finish<CurveG1>(fin, c, nwin, out_affine, out_inf);
}
template <>
inline void msm_host_finish_any<CurveG2>(const uint32_t* fin, int c, int nwin, uint32_t* out_affine, uint8_t* out_inf) {
  if (msm_host64_enabled() && !msm_finish_threads_enabled()) h64::msm_finish_serial<h64::Fp2>(fin, c, nwin, MSM_GROUP, msm_ngroups(c), 28, 24, out_affine, out_inf);
  else if (msm_host64_enabled()) h64::msm_finish<h64::Fp2>(fin, c, nwin, MSM_GROUP, msm_ngroups(c), 28, 24, out_affine, out_inf);
  else msm_host_finish<CurveG2>(fin, c, nwin, out_affine, out_inf);
}

}  // namespace ncg
