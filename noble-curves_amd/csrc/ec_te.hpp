// Twisted-Edwards a = -1 group law (ed25519), extended coordinates (X, Y, Z, T), T = XY/Z.
//
// Same formulas as the reference: dbl-2008-hwcd (src/abstract/edwards.ts:505-521) and
// add-2008-hwcd (:526-545), which are complete on ed25519 (a = -1 is a square, d is not), so
// no exceptional cases exist.  Table points are kept in "Niels" form (Y+X, Y-X, 2dT [, Z]) so a
// mixed add costs 7 field muls and a projective-table add 8 (the reference spends 9 + 1*d).
// The field is the radix-2^29 lazy form of fe9.hpp: sums and differences widen their limb bound, and
// the few that feed two products are weakly normalised once (`fe9_norm`).
#pragma once
#include "fp2.hpp"

namespace ncg {

template <class F>
struct EdExt {
  F X, Y, Z, T;
  static NCG_DI EdExt identity() { return {F::zero(), F::one(), F::one(), F::zero()}; }  // edwards.ts:370
};
template <class F>
struct EdNielsAff {  // affine point (x, y): (y+x, y-x, 2d*x*y)
  F yplusx, yminusx, t2d;
};
template <class F>
struct EdNielsProj {  // projective point: (Y+X, Y-X, Z, 2d*T)
  F yplusx, yminusx, Z, t2d;
};

// The ed25519 field is the radix-2^29 lazy form (fe9.hpp); every stored coordinate is a product or a
// normalised value, so the storage bound is 1 and sums / differences widen from there.
using FEd = Fe9<Fe9EdPR, 1>;

struct EdConsts {
  static NCG_DI FEd d() { return FEd::from_limbs(Fe9EdPR::D); }
  static NCG_DI FEd d2() { return FEd::from_limbs(Fe9EdPR::D2); }
  static NCG_DI FEd sqrt_m1() { return FEd::from_limbs(Fe9EdPR::SQRT_M1); }
};

// dbl-2008-hwcd with a = -1 (edwards.ts:505-521): E = (X+Y)^2 - A - B = 2XY is taken as one product
// (a square plus two bound-widening subtractions costs the same here), so 5M + 3S and two weak
// normalisations (F and H feed two products each).
template <class F>
NCG_DI EdExt<F> ed_dbl(const EdExt<F>& p) {
  auto A = f_sqr(p.X);
  auto B = f_sqr(p.Y);
  auto E = f_dbl(p.X * p.Y);                 // 2XY
  auto G = B - A;                            // D + B with D = a*A = -A
  auto H = fe9_norm(f_neg(A + B));           // D - B
  auto Fq = fe9_norm(G - f_dbl(f_sqr(p.Z)));  // G - C, C = 2 Z^2
  return {E * Fq, G * H, Fq * G, E * H};
}

// The same doubling without T3 (4M + 3S): valid when the next operation is another doubling,
// which never reads T.
template <class F>
NCG_DI EdExt<F> ed_dbl_no_t(const EdExt<F>& p) {
  auto A = f_sqr(p.X);
  auto B = f_sqr(p.Y);
  auto E = f_dbl(p.X * p.Y);
  auto G = B - A;
  auto H = fe9_norm(f_neg(A + B));
  auto Fq = fe9_norm(G - f_dbl(f_sqr(p.Z)));
  return {E * Fq, G * H, Fq * G, p.T};
}

// Niels forms are stored normalised (bound 1): the conversions below narrow, i.e. normalise once per entry,
// so that the additions can multiply them against the widened (Y - X), (Y + X) without more work.
template <class F>
NCG_DI EdNielsProj<F> ed_to_niels(const EdExt<F>& p, const F& d2) {
  return {p.Y + p.X, p.Y - p.X, p.Z, p.T * d2};
}
template <class F>
NCG_DI EdNielsAff<F> ed_affine_to_niels(const F& x, const F& y, const F& d2) {
  return {y + x, y - x, x * y * d2};
}

// extended + projective Niels (add-2008-hwcd-3 shape, a = -1): 8M.  neg: add -q.
template <class F>
NCG_DI EdExt<F> ed_add_niels(const EdExt<F>& p, const EdNielsProj<F>& q, bool neg) {
  F qa = neg ? q.yminusx : q.yplusx;
  F qb = neg ? q.yplusx : q.yminusx;
  auto A = (p.Y - p.X) * qb;
  auto B = (p.Y + p.X) * qa;
  auto C = f_cneg(p.T * q.t2d, neg);
  auto D = f_dbl(p.Z * q.Z);
  auto E = B - A;
  auto Fq = fe9_norm(D - C);
  auto G = fe9_norm(D + C);
  auto H = B + A;
  return {E * Fq, G * H, Fq * G, E * H};
}
// extended + affine Niels: 7M
template <class F>
NCG_DI EdExt<F> ed_madd_niels(const EdExt<F>& p, const EdNielsAff<F>& q, bool neg) {
  F qa = neg ? q.yminusx : q.yplusx;
  F qb = neg ? q.yplusx : q.yminusx;
  auto A = (p.Y - p.X) * qb;
  auto B = (p.Y + p.X) * qa;
  auto C = f_cneg(p.T * q.t2d, neg);
  auto D = f_dbl(p.Z);
  auto E = B - A;
  auto Fq = fe9_norm(D - C);
  auto G = fe9_norm(D + C);
  auto H = B + A;
  return {E * Fq, G * H, Fq * G, E * H};
}

// identity test on a projective representative: X == 0 and Y == Z (edwards.ts:482-495 vs ZERO)
template <class F>
NCG_DI bool ed_is_identity(const EdExt<F>& p) {
  return f_eqz(p.X) && f_eq(p.Y, p.Z);
}

// x^((p-5)/8) for p = 2^255 - 19: the reference's addition chain ed25519_pow_2_252_3
// (src/ed25519.ts:67-86): 250 squarings + 11 multiplications.
NCG_DI FEd ed_pow_p58(const FEd& x) {
  FEd x2 = f_sqr(x);
  FEd b2 = x2 * x;
  FEd b4 = fe9_sqr_n(b2, 2) * b2;
  FEd b5 = fe9_sqr_n(b4, 1) * x;
  FEd b10 = fe9_sqr_n(b5, 5) * b5;
  FEd b20 = fe9_sqr_n(b10, 10) * b10;
  FEd b40 = fe9_sqr_n(b20, 20) * b20;
  FEd b80 = fe9_sqr_n(b40, 40) * b40;
  FEd b160 = fe9_sqr_n(b80, 80) * b80;
  FEd b240 = fe9_sqr_n(b160, 80) * b80;
  FEd b250 = fe9_sqr_n(b240, 10) * b10;
  return fe9_sqr_n(b250, 2) * x;
}

// parity of the canonical residue (isNegativeLE, modular.ts:422)
template <int A>
NCG_DI bool ed_is_odd(const Fe9<Fe9EdPR, A>& x) {
  uint32_t c[9];
  fe9_canon_limbs<Fe9EdPR, A>(c, x);
  return (c[0] & 1u) != 0;
}

// sqrt(u/v) with the reference's three-candidate check (src/ed25519.ts:107-125); returns the
// root made non-negative (even) like the reference.
NCG_DI bool ed_uv_ratio(const FEd& u, const FEd& v, FEd& x_out) {
  FEd v3 = f_sqr(v) * v;
  FEd v7 = f_sqr(v3) * v;
  FEd pw = ed_pow_p58(u * v7);
  FEd x = u * v3 * pw;
  FEd vx2 = v * f_sqr(x);
  FEd root2 = x * EdConsts::sqrt_m1();
  auto negu = f_neg(u);
  bool useRoot1 = f_eq(vx2, u);
  bool useRoot2 = f_eq(vx2, negu);
  bool noRoot = f_eq(vx2, negu * EdConsts::sqrt_m1());
  if (useRoot2 || noRoot) x = root2;
  if (ed_is_odd(x)) x = f_neg(x);
  x_out = x;
  return useRoot1 || useRoot2;
}

// Point.fromBytes (src/abstract/edwards.ts:405-436) on 8 LE words.  Returns validity;
// x, y are affine coordinates (y taken mod p: ZIP-215 accepts y >= p).
NCG_DI bool ed_decompress(const uint32_t (&w)[8], bool zip215, FEd& x, FEd& y) {
  uint32_t yr[8];
#pragma unroll
  for (int i = 0; i < 8; i++) yr[i] = w[i];
  const bool sign = (yr[7] >> 31) != 0;
  yr[7] &= 0x7fffffffu;
  // strict mode: 0 <= y < p; zip215: y < 2^256 (always true once bit 255 is cleared)
  bool canonical;
  {
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) (void)__builtin_subc(yr[i], (uint32_t)ParamsEdP::P[i], bw, &bw);
    canonical = bw != 0;  // y < p
  }
  bool ok = zip215 || canonical;
  y = fe9_from_wire<Fe9EdPR>(yr);  // below 2^255: congruent to y mod p
  FEd y2 = f_sqr(y);
  FEd u = y2 - FEd::one();
  FEd v = EdConsts::d() * y2 + FEd::one();  // d*y^2 - a, a = -1
  FEd xx;
  bool valid = ed_uv_ratio(u, v, xx);
  ok = ok && valid;
  bool x_is0 = f_eqz(xx);
  if (!zip215 && x_is0 && sign) ok = false;
  // xx is even here; the sign bit asks for the odd root
  if (sign && !x_is0) xx = f_neg(xx);
  x = xx;
  return ok;
}

}  // namespace ncg
