// Twisted-Edwards a = -1 group law (ed25519), extended coordinates (X, Y, Z, T), T = XY/Z.
//
// Same formulas as the reference: dbl-2008-hwcd (src/abstract/edwards.ts:505-521) and
// add-2008-hwcd (:526-545), which are complete on ed25519 (a = -1 is a square, d is not), so
// no exceptional cases exist.  Table points are kept in "Niels" form (Y+X, Y-X, 2dT [, Z]) so a
// mixed add costs 7 field muls and a projective-table add 8 (the reference spends 9 + 1*d).
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

struct EdConsts {
  static NCG_DI FpEd d() { return FpEd::from_const(ParamsEdP::D); }
  static NCG_DI FpEd d2() { return FpEd::from_const(ParamsEdP::D2); }
  static NCG_DI FpEd sqrt_m1() { return FpEd::from_const(ParamsEdP::SQRT_M1); }
};

// dbl-2008-hwcd with a = -1: 4S + 4M  (edwards.ts:505-521)
template <class F>
NCG_DI EdExt<F> ed_dbl(const EdExt<F>& p) {
  F A = f_sqr(p.X);
  F B = f_sqr(p.Y);
  F C = f_dbl(f_sqr(p.Z));
  F D = f_neg(A);                      // a*A
  F E = f_sqr(p.X + p.Y) - A - B;
  F G = D + B;
  F Fq = G - C;
  F H = D - B;
  return {E * Fq, G * H, Fq * G, E * H};
}

// The same doubling without T3 (3M + 4S): valid when the next operation is another doubling,
// which never reads T.
template <class F>
NCG_DI EdExt<F> ed_dbl_no_t(const EdExt<F>& p) {
  F A = f_sqr(p.X);
  F B = f_sqr(p.Y);
  F C = f_dbl(f_sqr(p.Z));
  F D = f_neg(A);
  F E = f_sqr(p.X + p.Y) - A - B;
  F G = D + B;
  F Fq = G - C;
  F H = D - B;
  return {E * Fq, G * H, Fq * G, p.T};
}

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
  F A = (p.Y - p.X) * qb;
  F B = (p.Y + p.X) * qa;
  F C = p.T * q.t2d;
  if (neg) C = f_neg(C);
  F D = f_dbl(p.Z * q.Z);
  F E = B - A;
  F Fq = D - C;
  F G = D + C;
  F H = B + A;
  return {E * Fq, G * H, Fq * G, E * H};
}
// extended + affine Niels: 7M
template <class F>
NCG_DI EdExt<F> ed_madd_niels(const EdExt<F>& p, const EdNielsAff<F>& q, bool neg) {
  F qa = neg ? q.yminusx : q.yplusx;
  F qb = neg ? q.yplusx : q.yminusx;
  F A = (p.Y - p.X) * qb;
  F B = (p.Y + p.X) * qa;
  F C = p.T * q.t2d;
  if (neg) C = f_neg(C);
  F D = f_dbl(p.Z);
  F E = B - A;
  F Fq = D - C;
  F G = D + C;
  F H = B + A;
  return {E * Fq, G * H, Fq * G, E * H};
}

// identity test on a projective representative: X == 0 and Y == Z (edwards.ts:482-495 vs ZERO)
template <class F>
NCG_DI bool ed_is_identity(const EdExt<F>& p) {
  return p.X.is_zero() && (p.Y == p.Z);
}

// x^((p-5)/8) for p = 2^255 - 19: the reference's addition chain ed25519_pow_2_252_3
// (src/ed25519.ts:67-86): 250 squarings + 11 multiplications.
NCG_DI FpEd ed_pow_p58(const FpEd& x) {
  using PR = ParamsEdP;
  FpEd x2 = fp_sqr<PR>(x);
  FpEd b2 = x2 * x;
  FpEd b4 = fp_sqr_n<PR>(b2, 2) * b2;
  FpEd b5 = fp_sqr_n<PR>(b4, 1) * x;
  FpEd b10 = fp_sqr_n<PR>(b5, 5) * b5;
  FpEd b20 = fp_sqr_n<PR>(b10, 10) * b10;
  FpEd b40 = fp_sqr_n<PR>(b20, 20) * b20;
  FpEd b80 = fp_sqr_n<PR>(b40, 40) * b40;
  FpEd b160 = fp_sqr_n<PR>(b80, 80) * b80;
  FpEd b240 = fp_sqr_n<PR>(b160, 80) * b80;
  FpEd b250 = fp_sqr_n<PR>(b240, 10) * b10;
  return fp_sqr_n<PR>(b250, 2) * x;
}

// sqrt(u/v) with the reference's three-candidate check (src/ed25519.ts:107-125); returns the
// root made non-negative (even) like the reference.  Values are in Montgomery form.
NCG_DI bool ed_uv_ratio(const FpEd& u, const FpEd& v, FpEd& x_out) {
  using PR = ParamsEdP;
  FpEd v3 = fp_sqr<PR>(v) * v;
  FpEd v7 = fp_sqr<PR>(v3) * v;
  FpEd pw = ed_pow_p58(u * v7);
  FpEd x = u * v3 * pw;
  FpEd vx2 = v * fp_sqr<PR>(x);
  FpEd root2 = x * EdConsts::sqrt_m1();
  FpEd negu = fp_neg<PR>(u);
  bool useRoot1 = vx2 == u;
  bool useRoot2 = vx2 == negu;
  bool noRoot = vx2 == negu * EdConsts::sqrt_m1();
  if (useRoot2 || noRoot) x = root2;
  // isNegativeLE: parity of the canonical residue (modular.ts:422)
  FpEd xc = fp_from_mont<PR>(x);
  if (xc.v[0] & 1u) x = fp_neg<PR>(x);
  x_out = x;
  return useRoot1 || useRoot2;
}

// Point.fromBytes (src/abstract/edwards.ts:405-436) on 8 LE words.  Returns validity;
// x, y are Montgomery-form affine coordinates (y reduced mod p: ZIP-215 accepts y >= p).
NCG_DI bool ed_decompress(const uint32_t (&w)[8], bool zip215, FpEd& x, FpEd& y) {
  using PR = ParamsEdP;
  FpEd yr;
#pragma unroll
  for (int i = 0; i < 8; i++) yr.v[i] = w[i];
  const bool sign = (yr.v[7] >> 31) != 0;
  yr.v[7] &= 0x7fffffffu;
  // strict mode: 0 <= y < p; zip215: y < 2^256 (always true once bit 255 is cleared)
  bool canonical;
  {
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) (void)__builtin_subc(yr.v[i], (uint32_t)PR::P[i], bw, &bw);
    canonical = bw != 0;  // y < p
  }
  bool ok = zip215 || canonical;
  y = fp_to_mont<PR>(yr);  // reduces mod p
  FpEd y2 = fp_sqr<PR>(y);
  FpEd u = y2 - FpEd::one();
  FpEd v = EdConsts::d() * y2 + FpEd::one();  // d*y^2 - a, a = -1
  FpEd xx;
  bool valid = ed_uv_ratio(u, v, xx);
  ok = ok && valid;
  bool x_is0 = xx.is_zero();
  if (!zip215 && x_is0 && sign) ok = false;
  // xx is even here; the sign bit asks for the odd root
  if (sign && !x_is0) xx = fp_neg<PR>(xx);
  x = xx;
  return ok;
}

}  // namespace ncg
