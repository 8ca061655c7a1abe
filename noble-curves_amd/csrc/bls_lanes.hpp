// Lane-level helpers shared by the bls12-381 decoders (decode.hip) and the hash-to-curve maps
// (h2c.hip): constants, fixed exponentiations, the BLS-parameter ladder and the Fp2 square root.
#pragma once
#include "curves.hpp"

namespace ncg {

// a^((p - 3) / 4): the shared power of the 3 mod 4 square roots (modular.ts:205-215 sqrt3mod4 value; the
// Fp2 root of bls_lanes / h2c runs two of them).  Sliding windows over the compile-time schedule of fe29.hpp.
NCG_DI Fe29<2> fe29_pow_sqrt_m1(const Fe29<2>& a) { return fe29_pow_sched(a, BlsPowSched<NCG_POW_W>::SQRT_M1); }
// a^((p + 1) / 4): the candidate root itself (the G1 decoder)
NCG_DI Fe29<2> fe29_pow_sqrt(const Fe29<2>& a) { return fe29_pow_sched(a, BlsPowSched<NCG_POW_W>::SQRT); }

// [x]P for the BLS parameter x = 0xd201000000010000 (Jacobian double-and-add; x is public and
// identical for every lane)
template <class F>
NCG_DI Jac<F> bls_mul_by_x(const Jac<F>& p) {
  const uint64_t X = 0xD201000000010000ull;
  Jac<F> r = p;  // top bit
  for (int bit = 62; bit >= 0; bit--) {
    r = jac_dbl(r);
    if ((X >> bit) & 1ull) r = jac_add(r, p);
  }
  return r;
}

NCG_DI Fe29<1> fe29_const(const uint32_t (&c)[14]) {
  Fe29<1> r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = c[i];
  return r;
}
NCG_DI void be48_to_words(const uint8_t* __restrict__ in, uint32_t (&w)[12]) {
#pragma unroll
  for (int i = 0; i < 12; i++) {
    const uint8_t* b = in + (11 - i) * 4;
    w[i] = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3];
  }
}
NCG_DI bool words12_lt_p(const uint32_t (&w)[12]) {
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) (void)__builtin_subc(w[i], (uint32_t)BlsFpConsts::P32[i], bw, &bw);
  return bw != 0;
}
NCG_DI bool words12_gt_half_p(const uint32_t (&w)[12]) {  // (2 w) / p != 0
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) (void)__builtin_subc((uint32_t)BlsFpConsts::HALF_P[i], w[i], bw, &bw);
  return bw != 0;
}
NCG_DI void words12_neg_mod_p(uint32_t (&w)[12]) {  // w = p - w for w != 0
  uint32_t any = 0, bw = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) any |= w[i];
  if (any == 0) return;
#pragma unroll
  for (int i = 0; i < 12; i++) w[i] = __builtin_subc((uint32_t)BlsFpConsts::P32[i], w[i], bw, &bw);
}

// ---- lane-paired Fp2 helpers (CurveG2P: one Fp2 element per lane pair, fe29.hpp) shared by the G2 decoder's
// subgroup stage and the hash-to-curve cofactor stage
NCG_DI Fe29x2P<1> p2_const(const uint32_t (&c0)[14], const uint32_t (&c1)[14]) {
  return Fe29x2P<1>(fe29_select(pair_odd(), fe29_const(c1), fe29_const(c0)));
}
NCG_DI FeBls2P p2_conj(const FeBls2P& a) {  // c0 - c1 u: the odd lane negates its half
  return FeBls2P(fe29_select(pair_odd(), f_neg(a.h), a.h));
}
NCG_DI Jac<FeBls2P> g2p_psi(const Jac<FeBls2P>& P) {
  if (P.is_inf()) return P;
  const Fe29x2P<1> psx = p2_const(ParamsBls29::PSI_X_C0, ParamsBls29::PSI_X_C1);
  const Fe29x2P<1> psy = p2_const(ParamsBls29::PSI_Y_C0, ParamsBls29::PSI_Y_C1);
  return {p2_conj(P.X) * psx, p2_conj(P.Y) * psy, p2_conj(P.Z)};
}

// Square root in Fp2 = Fp[u]/(u^2+1), p = 3 mod 4.  The reference's complex method
// (tower.ts:476-500) spends sqrt(norm) + Legendre(d) + sqrt(d) + one inversion; here the Legendre
// symbol, the root and the inverse all come out of ONE power t = d^((p-3)/4):
//   s = t d satisfies s^2 = +-d and t s = d^((p-1)/2) = +-1, so 1/s = +-t;
//   s^2 =  d: root (s, c1/(2s));   s^2 = -d (d a non-residue): root (c1/(2s), s), using
//   d d' = -c1^2/4 for the reference's second candidate d' = d - a.
// Which of the two roots comes out is irrelevant to the caller (the sort bit picks the sign);
// existence is decided exactly like the reference: norm must be a square and root^2 == num.
NCG_DI bool fe29x2_sqrt(const Fe29x2<2>& num, Fe29x2<2>& root) {
  const Fe29<1> half = fe29_const(ParamsBls29::HALF);
  Fe29<2> norm = (f_sqr(num.c0) + f_sqr(num.c1)) * Fe29<1>::one();
  Fe29<2> a = fe29_pow_sqrt_m1(norm) * norm;
  bool ok = f_eq(f_sqr(a), norm);
  const bool c1_zero = f_eqz(num.c1);
  Fe29<2> d = (a + num.c0) * half;
  if (c1_zero) d = num.c0;
  Fe29<2> t = fe29_pow_sqrt_m1(d);
  Fe29<2> s = t * d;
  const bool residue = f_eq(f_sqr(s), d);
  Fe29<2> o = num.c1 * half * t;  // c1 / (2 s) up to the sign fixed below
  if (residue) {
    root = {s, o};
  } else {
    root = {f_neg(o) * Fe29<1>::one(), s};
  }
  Fe29x2<2> chk = f_sqr(root);
  return ok && f_eq(chk.c0, num.c0) && f_eq(chk.c1, num.c1);
}

}  // namespace ncg
