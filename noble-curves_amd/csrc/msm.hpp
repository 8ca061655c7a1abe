// Pippenger multi-scalar multiplication sum_i s_i * P_i on one GPU.
//
// Reproduces the result of the reference's `pippenger(c, points, scalars)`
// (src/abstract/curve.ts:863-905).  The reference walks unsigned c-bit windows serially,
// adds every point into `buckets[digit]` with complete projective additions (:890-894),
// folds the buckets with a running sum (:897-900) and Horner-combines windows (:901-902).
// Only the final group element is contractual (canonical affine), so the GPU
// pipeline is organised for the machine instead:
//
//   1. k_points_to_mont   affine wire points -> Montgomery limbs, once            (HBM stream)
//   2. k_msm_digits       signed c-bit digits for every (scalar, window): add the constant
//                         H' = sum 2^(c-1) 2^(cw) once, then every window is an independent
//                         bit-field minus 2^(c-1)  (digits in [-2^(c-1), 2^(c-1)-1]; signed
//                         digits halve the bucket count - an integer identity, always exact)
//   3. k_msm_hist / k_msm_scan / k_msm_scatter   counting sort of point indices by
//                         (window, |digit|): per-block LDS histograms (2^(c-1) counters, 128 KB
//                         at c = 16), one scan block per window, LDS-atomic scatter
//   4. k_msm_accum        bucket sums: XYZZ accumulator += affine point (mixed add, 10 field
//                         muls instead of the reference's 14), points gathered by sorted index
//   5. k_msm_reduce_level bucket fold sum_b b*B_b as log2(#buckets) pairwise levels
//                         (S' = S[2q]+S[2q+1], R = S[2q+1]; pending R-arrays are pair-summed),
//                         depth log instead of the reference's serial running sum
//   6. host               the c*nwin surviving points are Horner-combined (256 doublings: a
//                         serial chain that a latency-optimised core does ~30x faster than one
//                         GPU lane) and normalised to affine (curve.ts:311-326 / toAffine)
//
// Zero digits are skipped (adding into bucket 0 never reaches the reference's result either,
// curve.ts:896 "Skip first bucket"), infinity inputs are skipped, P = Q / P = -Q inside a
// bucket are handled by the group-law routines (ec_sw.hpp).
#pragma once
#include "curves.hpp"

namespace ncg {

struct MsmPlan {
  int n = 0;       // points
  int c = 0;       // window bits
  int nwin = 0;    // windows
  int nb = 0;      // buckets per window = 2^(c-1)
  int Q = 0;       // sort chunks (blocks per window)
  int chunk = 0;   // points per chunk
  uint32_t hconst[10];  // H' = sum_w 2^(c-1) * 2^(c w), 10 LE limbs
};

// sizes in 32-bit words of one stored point
template <class C> struct MsmSizes {
  using F = typename C::F;
  static constexpr int FW = FieldIO<F>::WORDS;        // stored words per field element
  static constexpr int WIRE_AFF = 2 * FieldWire<F>::WORDS;  // wire words per affine point
  static constexpr int AFF = 2 * FW;
  static constexpr int XYZZ = 4 * FW;
};

template <class F>
NCG_DI Xyzz<F> xyzz_load(const uint32_t* p) {
  constexpr int FW = FieldIO<F>::WORDS;
  return {FieldIO<F>::load(p), FieldIO<F>::load(p + FW), FieldIO<F>::load(p + 2 * FW), FieldIO<F>::load(p + 3 * FW)};
}
template <class F>
NCG_DI void xyzz_store(uint32_t* p, const Xyzz<F>& a) {
  constexpr int FW = FieldIO<F>::WORDS;
  FieldIO<F>::store(p, a.X);
  FieldIO<F>::store(p + FW, a.Y);
  FieldIO<F>::store(p + 2 * FW, a.ZZ);
  FieldIO<F>::store(p + 3 * FW, a.ZZZ);
}

}  // namespace ncg
