// bls12-381 G1 batch multiply for points KNOWN to lie in the prime-order subgroup (resident sets that passed or
// were decoded with the reference's isTorsionFree): the GLV ladder of mulvar.hpp on CurveG1E (curves.hpp) -
// k = k1 + k2 z^2 with 128-bit halves, 33 windows x (4 doublings + 2 additions) instead of 65 x (4 + 1).
// Same table build, same small-order fallback and the same outputs as the generic kernel.
#include "mulvar.hpp"
#include "bls_lanes.hpp"
#include "host_api.hpp"

namespace ncg {

hipError_t mul_var_batch_g1_subgroup(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                                     uint32_t* jac_tmp, hipStream_t st) {
  return launch_mul_var_gtab<CurveG1E, 4, 2>(pts, scalars, out, out_inf, n, jac_tmp, st);
}

// ------------------------------------------------------------------------------------------------ G2
// bls12-381 G2 batch multiply on a verified set: psi acts as -z on the subgroup (endo.hpp; the identity of the
// reference's own subgroup test, bls12-381.ts:599-601), so k = d0 + d1 z + d2 z^2 + d3 z^3 with |d_e| < 2^63 and
//     k P = d0 P + d1 (-psi P) + d2 (psi^2 P) + d3 (-psi^3 P):
// 16 windows x (4 doublings + 4 additions) instead of 65 x (4 + 1).  One table of odd multiples T_j = (2j+1) P as
// in the generic kernel (affine points of an isomorphic curve, shared Z), with two additions to it:
//   * the shared Z is made REAL (every entry rescaled once more by conj(Z): Z conj(Z) is a norm), so that the
//     conjugation inside psi commutes with the isomorphism and psi(T_j) = (conj(x_j) PSI_X, conj(y_j) PSI_Y) is a
//     point of the same isomorphic curve;
//   * those images are stored behind the table (the psi^2 twist of either table is one Fp product and a sign).
// Streams 0 / 2 read the table, streams 1 / 3 its psi image; 2 and 3 multiply x by PSI2_X; the y signs follow
// k_points_endo_g2 (msm_endo.hip).  Scalars >= r and small-order points (table build degenerate) take the complete
// ladder, as in the generic kernel.
template <int W>
struct G2PsiCfg {
  using F = CurveG2P::F;
  static constexpr int FW = FieldIO<F>::WORDS, TW = FieldIO<F>::LANE_WORDS, WW = FieldWire<F>::WORDS;
  static constexpr int TS = 1 << (W - 1);
  static constexpr int M = (64 + W - 1) / W;     // |d_e| + 1 < 2^64
  static constexpr int ZR_OFF = TS * 2 * TW;     // Z ratios of the build
  static constexpr int PSI_OFF = TS * 3 * TW;    // psi images (x, y)
  static constexpr int TAB_WORDS = TS * 5 * TW;  // per lane
};

template <int W>
NCG_DI void mul_var_lane_g2psi(const uint32_t* __restrict__ pt_wire, const uint32_t* __restrict__ k_wire,
                               uint32_t* __restrict__ out_jac, bool active, uint32_t* __restrict__ tab) {
  using Cfg = G2PsiCfg<W>;
  using C = CurveG2P;
  using F = typename C::F;
  constexpr int FW = Cfg::FW, TW = Cfg::TW, TS = Cfg::TS, M = Cfg::M;
  Affine<F> P = load_affine_wire<F>(pt_wire);
  uint32_t k[8];
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = k_wire[i];
  const bool trivial_zero = P.is_inf() || mp_is_zero<8>(k);
  bool degenerate = false;
  {
    static constexpr uint32_t BLS_R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                          0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    uint32_t bw = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) (void)__builtin_subc(k[j], (uint32_t)BLS_R[j], bw, &bw);
    degenerate = bw == 0;  // k >= r: outside the split's range
  }
  // ---- table of odd multiples (mul_var_lane's build with the Z ratios in memory), then the real shared Z
  F Zr;
  {
    Jac<F> D = jac_dbl(Jac<F>{P.x, P.y, F::one()});
    auto dz2 = f_sqr(D.Z);
    auto dz3 = dz2 * D.Z;
    Affine<F> Dc{D.X, D.Y};   // co-Z steps (mulvar.hpp coz_addu); the running Z is kept here: the rescale below starts from cz
    Affine<F> T{P.x * dz2, P.y * dz3};
    F Tz = F::one();
    FieldIO<F>::store_strided(tab, 1, T.x);
    FieldIO<F>::store_strided(tab + TW, 1, T.y);
#pragma unroll 1
    for (int j = 1; j < TS; j++) {
      F zj;
      coz_addu(Dc, T, zj, degenerate);
      Tz = Tz * zj;
      FieldIO<F>::store_strided(tab + Cfg::ZR_OFF + j * TW, 1, zj);
      FieldIO<F>::store_strided(tab + j * 2 * TW, 1, T.x);
      FieldIO<F>::store_strided(tab + j * 2 * TW + TW, 1, T.y);
    }
    const F Zg = D.Z * Tz;
    const F cz = p2_conj(Zg);
    Zr = Zg * cz;
    const Fe29x2P<1> psx = p2_const(ParamsBls29::PSI_X_C0, ParamsBls29::PSI_X_C1);
    const Fe29x2P<1> psy = p2_const(ParamsBls29::PSI_Y_C0, ParamsBls29::PSI_Y_C1);
    F s = cz;
#pragma unroll 1
    for (int j = TS - 1; j >= 0; j--) {
      if (j < TS - 1) s = s * FieldIO<F>::load_strided(tab + Cfg::ZR_OFF + (j + 1) * TW, 1);
      auto s2 = f_sqr(s);
      auto s3 = s2 * s;
      const F x = FieldIO<F>::load_strided(tab + j * 2 * TW, 1) * s2;
      const F y = FieldIO<F>::load_strided(tab + j * 2 * TW + TW, 1) * s3;
      FieldIO<F>::store_strided(tab + j * 2 * TW, 1, x);
      FieldIO<F>::store_strided(tab + j * 2 * TW + TW, 1, y);
      const F px = p2_conj(x) * psx;
      const F py = p2_conj(y) * psy;
      FieldIO<F>::store_strided(tab + Cfg::PSI_OFF + j * 2 * TW, 1, px);
      FieldIO<F>::store_strided(tab + Cfg::PSI_OFF + j * 2 * TW + TW, 1, py);
    }
  }
  // ---- scalar recoding: four balanced sub-scalars, sign + magnitude
  SignedOddWindows<3, W, M> win[4];
  bool neg[4];
  {
    uint32_t sub[4][6];
    bls_endo_split4(sub, k);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      neg[e] = (sub[e][5] >> 31) != 0;
      if (neg[e]) mp_neg<6>(sub[e]);
      const uint32_t m3[3] = {sub[e][0], sub[e][1], sub[e][2]};
      win[e].template init<3>(m3);
    }
  }
  const Fe29<1> k2 = fe29_const(BlsH2c::PSI2_X);
  // stream e adds (sign) * z^e T_j:  z T = (psi x, -psi y), z^2 T = (k2 x, -y), z^3 T = (k2 psi x, psi y)
  auto add_stream = [&](Jac<F> R, int e, int idx, bool minus) -> Jac<F> {
    const uint32_t* base = tab + ((e & 1) ? Cfg::PSI_OFF : 0) + idx * 2 * TW;
    const F qx = FieldIO<F>::load_strided(base, 1);
    const F qy = FieldIO<F>::load_strided(base + TW, 1);
    const bool flip = minus != (e == 1 || e == 2);
    if (e >= 2) return jac_madd_q(R, Fe29x2P<2>(qx.h * k2), f_cneg(qy, flip));
    return jac_madd_q(R, qx, f_cneg(qy, flip));
  };
  Jac<F> R = Jac<F>::inf();
#pragma unroll 1
  for (int i = 0; i < M; i++) {
    if (i > 0) {
#pragma unroll 1
      for (int d = 0; d < W; d++) R = jac_dbl(R);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int d = win[e].pop();
      R = add_stream(R, e, ((d < 0 ? -d : d) - 1) >> 1, (d < 0) != neg[e]);
    }
  }
  // even sub-scalars were bumped by one: take the extra point back out
#pragma unroll
  for (int e = 0; e < 4; e++) {
    if (win[e].was_even) R = add_stream(R, e, 0, !neg[e]);
  }
  const bool ladder_inf = R.is_inf();
  R.Z = R.Z * Zr;
  if (ladder_inf) R = Jac<F>::inf();
  if (degenerate) R = mul_var_slow<C>(pt_wire, k_wire);
  if (trivial_zero || R.is_inf()) R = Jac<F>::inf();
  if (active) {
    FieldIO<F>::store(out_jac, R.X);
    FieldIO<F>::store(out_jac + FW, R.Y);
    FieldIO<F>::store(out_jac + 2 * FW, R.Z);
  }
}

template <int W, int MINW>
__global__ void __launch_bounds__(64, MINW)
k_mul_var_g2psi(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ scalars, uint32_t* __restrict__ out_jac,
                uint32_t* __restrict__ gtab, int n) {
  using Cfg = G2PsiCfg<W>;
  const int lane_idx = blockIdx.x * 64 + threadIdx.x;  // one table per LANE of the pair
  const int idx = lane_idx >> 1;
  const bool active = idx < n;
  const int src = active ? idx : n - 1;
  mul_var_lane_g2psi<W>(pts + (size_t)src * 2 * Cfg::WW, scalars + (size_t)src * 8, out_jac + (size_t)src * 3 * Cfg::FW, active,
                        gtab + (size_t)lane_idx * Cfg::TAB_WORDS);
}

size_t mul_var_g2_subgroup_tmp_bytes(int n) {
  using Cfg = G2PsiCfg<4>;
  return pad64(n) * (3 * (size_t)Cfg::FW + 2 * (size_t)Cfg::TAB_WORDS) * 4;
}

hipError_t mul_var_batch_g2_subgroup(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                                     uint32_t* jac_tmp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  using Cfg = G2PsiCfg<4>;
  constexpr int K = 4;
  uint32_t* gtab = jac_tmp + pad64(n) * 3 * Cfg::FW;
  const unsigned blocks = (unsigned)((((size_t)n << 1) + 63) / 64);
  hipLaunchKernelGGL((k_mul_var_g2psi<4, 2>), dim3(blocks), dim3(64), 0, st, pts, scalars, jac_tmp, gtab, n);
  const int threads = ((n + K - 1) / K) << 1;
  hipLaunchKernelGGL((k_jac_batch_affine<CurveG2P, K>), dim3((threads + 255) / 256), dim3(256), 0, st, jac_tmp, out, out_inf, n);
  return hipGetLastError();
}

}  // namespace ncg
