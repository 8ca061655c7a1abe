// Short-Weierstrass a = 0 group law for secp256k1, bls12-381 G1 (Fp) and G2 (Fp2).
//
// The reference uses the complete projective Renes-Costello-Batina formulas
// (src/abstract/weierstrass.ts:793-828 double, :834-880 add; 13-14 field muls each).
// Only the *group element* is contractual (canonical affine equality), so the
// device uses cheaper coordinates and handles the exceptional cases explicitly:
//   * Jacobian (X, Y, Z), x = X/Z^2, y = Y/Z^3, infinity Z = 0     - scalar-mult ladders
//       dbl-2009-l (2M+5S), madd-2007-bl-style mixed add (7M+4S), add-2007-bl (11M+5S)
//   * XYZZ (X, Y, ZZ, ZZZ), x = X/ZZ, y = Y/ZZZ, infinity ZZ = 0   - MSM buckets
//       madd-2008-s (8M+2S), add-2008-s (12M+2S), mdbl-2008-s-1, dbl-2008-s-1
// P = Q, P = -Q and O are detected and routed to doubling / infinity, which is what the
// reference's complete formulas compute implicitly (weierstrass.ts:789-792, 830-833).
#pragma once
#include "fp2.hpp"

namespace ncg {

// a*b - c*d; a field with a fused multiply-accumulate (fe29.hpp) overloads it with a single reduction
template <class A, class B, class C, class D>
NCG_DI auto f_mulsub(const A& a, const B& b, const C& c, const D& d) -> decltype(a * b - c * d) {
  return a * b - c * d;
}


template <class F>
struct Affine {  // wire convention: infinity is (0, 0)  (weierstrass.ts:716, :966)
  F x, y;
  NCG_DI bool is_inf() const { return x.is_zero() && y.is_zero(); }
};

template <class F>
struct Jac {
  F X, Y, Z;
  static NCG_DI Jac inf() { return {F::one(), F::one(), F::zero()}; }
  NCG_DI bool is_inf() const { return Z.is_zero(); }
};

template <class F>
struct Xyzz {
  F X, Y, ZZ, ZZZ;
  static NCG_DI Xyzz inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
  NCG_DI bool is_inf() const { return ZZ.is_zero(); }
};

// ----------------------------------------------------------------- Jacobian
template <class F>
NCG_DI Jac<F> jac_from_affine(const Affine<F>& p) {
  if (p.is_inf()) return Jac<F>::inf();
  return {p.x, p.y, F::one()};
}

template <class F>
NCG_DI Jac<F> jac_neg(const Jac<F>& p) {
  return {p.X, f_neg(p.Y), p.Z};
}

// Fields whose products of a literal zero are literal zeros again (so "Z = 0 stays Z = 0" through
// a doubling for free).  False for the unpaired Fp2 over Fe29, whose Karatsuba subtractions turn
// 0 into a non-literal multiple of p: there the doubling returns infinity explicitly.
template <class F> struct KeepsLiteralZero { static constexpr bool value = true; };
template <int B> struct KeepsLiteralZero<Fe29x2<B>> { static constexpr bool value = false; };

// dbl-2009-l (a = 0): 2M + 5S.  Z = 0 stays Z = 0; no point of order 2 exists on these curves.
template <class F>
NCG_DI Jac<F> jac_dbl(const Jac<F>& p) {
  if constexpr (!KeepsLiteralZero<F>::value) {
    if (p.is_inf()) return Jac<F>::inf();
  }
  auto A = f_sqr(p.X);
  auto B = f_sqr(p.Y);
  auto C = f_sqr(B);
  auto t = f_sqr(p.X + B) - A - C;
  auto D = f_dbl(t);
  auto E = f_dbl(A) + A;
  auto Fq = f_sqr(E);
  auto X3 = Fq - f_dbl(D);
  auto C8 = f_dbl(f_dbl(f_dbl(C)));
  auto Y3 = E * (D - X3) - C8;
  auto Z3 = f_dbl(p.Y * p.Z);
  return {X3, Y3, Z3};
}

// Jacobian + affine (x, y) (madd-2007-bl without the 2x scaling): 8M + 3S, P = +-Q explicit.  x and y
// may be of a wider-bound type than the stored coordinates (a conditionally negated table entry).
template <class F, class FX, class FY>
NCG_DI Jac<F> jac_madd_xy(const Jac<F>& p, const FX& qx, const FY& qy) {
  auto Z1Z1 = f_sqr(p.Z);
  auto U2 = qx * Z1Z1;
  auto S2 = qy * p.Z * Z1Z1;
  auto H = U2 - p.X;
  auto R = S2 - p.Y;
  if (f_eqz(H)) {
    if (f_eqz(R)) return jac_dbl(p);  // P == Q
    return Jac<F>::inf();                // P == -Q
  }
  auto HH = f_sqr(H);
  auto HHH = H * HH;
  auto V = p.X * HH;
  auto X3 = f_sqr(R) - HHH - f_dbl(V);
  auto Y3 = R * (V - X3) - p.Y * HHH;
  auto Z3 = p.Z * H;
  return {X3, Y3, Z3};
}
// the same with the infinity cases of either operand (affine infinity = literal (0, 0))
template <class F, class FX, class FY>
NCG_DI Jac<F> jac_madd_q(const Jac<F>& p, const FX& qx, const FY& qy) {
  if (qx.is_zero() && qy.is_zero()) return p;
  if (p.is_inf()) return {qx, qy, F::one()};
  return jac_madd_xy(p, qx, qy);
}
template <class F>
NCG_DI Jac<F> jac_madd(const Jac<F>& p, const Affine<F>& q) {
  return jac_madd_q(p, q.x, q.y);
}

// Jacobian + Jacobian: 12M + 4S, exceptional cases explicit.
template <class F>
NCG_DI Jac<F> jac_add(const Jac<F>& p, const Jac<F>& q) {
  if (q.is_inf()) return p;
  if (p.is_inf()) return q;
  auto Z1Z1 = f_sqr(p.Z);
  auto Z2Z2 = f_sqr(q.Z);
  auto U1 = p.X * Z2Z2;
  auto U2 = q.X * Z1Z1;
  auto S1 = p.Y * q.Z * Z2Z2;
  auto S2 = q.Y * p.Z * Z1Z1;
  auto H = U2 - U1;
  auto R = S2 - S1;
  if (f_eqz(H)) {
    if (f_eqz(R)) return jac_dbl(p);
    return Jac<F>::inf();
  }
  auto HH = f_sqr(H);
  auto HHH = H * HH;
  auto V = U1 * HH;
  auto X3 = f_sqr(R) - HHH - f_dbl(V);
  auto Y3 = R * (V - X3) - S1 * HHH;
  auto Z3 = p.Z * q.Z * H;
  return {X3, Y3, Z3};
}

// Jacobian -> affine with a supplied inverse of Z (weierstrass.ts:951-969 toAffine(invZ)).
template <class F, class ZI>
NCG_DI Affine<F> jac_to_affine(const Jac<F>& p, const ZI& zinv) {
  if (p.is_inf()) return {F::zero(), F::zero()};
  auto zi2 = f_sqr(zinv);
  return {p.X * zi2, p.Y * zi2 * zinv};
}

// ----------------------------------------------------------------- Jacobian over Fe9 (secp256k1)
// Same formulas with the bookkeeping of the lazy limb bounds made explicit (fe9.hpp): a product wants
// bound(a)*bound(b) <= 7 and a stored coordinate bound <= 2, so the few differences that feed a product
// or a store are weakly normalised (`fe9_norm`, 3 plain ops per limb), and the doubling uses 4XB = 4*X*Y^2
// directly instead of the (X+B)^2 - A - C squaring trick (a multiply is cheaper than a square plus two
// normalised subtractions here).  Overloads: picked for F = Fe9<PR, B> by partial ordering.
template <class PR, int B>
NCG_DI Jac<Fe9<PR, B>> jac_dbl(const Jac<Fe9<PR, B>>& p) {
  auto A = f_sqr(p.X);
  auto Bq = f_sqr(p.Y);
  auto C8 = f_dbl(f_sqr(f_dbl(Bq)));                 // 8 Y^4, bound 2
  auto D = fe9_norm(f_dbl(f_dbl(p.X * Bq)));         // 4 X Y^2
  auto E = fe9_norm(A + A + A);
  auto X3 = f_sqr(E) - f_dbl(D);                     // bound 4
  auto Y3 = E * (D - X3) - C8;                       // (bound 6) * 1, then 1 - 2 -> 4
  auto Z3 = f_dbl(p.Y * p.Z);
  return {X3, Y3, Z3};
}

// Jacobian + affine (x, y) with y possibly a (conditionally) negated table entry of a wider bound.
template <class PR, int B, int BX, int BY>
NCG_DI Jac<Fe9<PR, B>> jac_madd_xy(const Jac<Fe9<PR, B>>& p, const Fe9<PR, BX>& qx, const Fe9<PR, BY>& qy) {
  using F = Fe9<PR, B>;
  auto Z1Z1 = f_sqr(p.Z);
  auto U2 = qx * Z1Z1;
  auto S2 = qy * (p.Z * Z1Z1);
  auto Hw = U2 - p.X;
  auto Rw = S2 - p.Y;
  if (f_eqz(Hw)) {
    if (f_eqz(Rw)) return jac_dbl(p);  // P == Q
    return Jac<F>::inf();              // P == -Q
  }
  auto H = fe9_norm(Hw);
  auto R = fe9_norm(Rw);
  auto HH = f_sqr(H);
  auto HHH = H * HH;
  auto V = p.X * HH;
  auto X3 = fe9_norm(f_sqr(R) - HHH - f_dbl(V));
  auto Y3 = R * (V - X3) - p.Y * HHH;
  auto Z3 = p.Z * H;
  return {X3, Y3, Z3};
}
template <class PR, int B>
NCG_DI Jac<Fe9<PR, B>> jac_add(const Jac<Fe9<PR, B>>& p, const Jac<Fe9<PR, B>>& q) {
  using F = Fe9<PR, B>;
  if (q.is_inf()) return p;
  if (p.is_inf()) return q;
  auto Z1Z1 = f_sqr(p.Z);
  auto Z2Z2 = f_sqr(q.Z);
  auto U1 = p.X * Z2Z2;
  auto U2 = q.X * Z1Z1;
  auto S1 = p.Y * (q.Z * Z2Z2);
  auto S2 = q.Y * (p.Z * Z1Z1);
  auto Hw = U2 - U1;
  auto Rw = S2 - S1;
  if (f_eqz(Hw)) {
    if (f_eqz(Rw)) return jac_dbl(p);
    return Jac<F>::inf();
  }
  auto H = fe9_norm(Hw);
  auto R = fe9_norm(Rw);
  auto HH = f_sqr(H);
  auto HHH = H * HH;
  auto V = U1 * HH;
  auto X3 = fe9_norm(f_sqr(R) - HHH - f_dbl(V));
  auto Y3 = R * (V - X3) - S1 * HHH;
  auto Z3 = (p.Z * q.Z) * H;
  return {X3, Y3, Z3};
}

// ----------------------------------------------------------------- XYZZ
template <class F>
NCG_DI Xyzz<F> xyzz_from_affine(const Affine<F>& p) {
  if (p.is_inf()) return Xyzz<F>::inf();
  return {p.x, p.y, F::one(), F::one()};
}

// mdbl-2008-s-1 (a = 0): double an affine point into XYZZ.
template <class F>
NCG_DI Xyzz<F> xyzz_mdbl(const Affine<F>& p) {
  auto U = f_dbl(p.y);
  auto V = f_sqr(U);
  auto W = U * V;
  auto S = p.x * V;
  auto xx = f_sqr(p.x);
  auto M = f_dbl(xx) + xx;
  auto X3 = f_sqr(M) - f_dbl(S);
  auto Y3 = f_mulsub(M, S - X3, W, p.y);
  return {X3, Y3, V, W};
}

// dbl-2008-s-1 (a = 0)
template <class F>
NCG_DI Xyzz<F> xyzz_dbl(const Xyzz<F>& p) {
  if (p.is_inf()) return p;
  auto U = f_dbl(p.Y);
  auto V = f_sqr(U);
  auto W = U * V;
  auto S = p.X * V;
  auto xx = f_sqr(p.X);
  auto M = f_dbl(xx) + xx;
  auto X3 = f_sqr(M) - f_dbl(S);
  auto Y3 = f_mulsub(M, S - X3, W, p.Y);
  return {X3, Y3, V * p.ZZ, W * p.ZZZ};
}

// madd-2008-s: XYZZ + affine, 8M + 2S; `neg` adds -q instead.
template <class F>
NCG_DI Xyzz<F> xyzz_madd(const Xyzz<F>& p, const Affine<F>& q_in, bool neg = false) {
  if (q_in.is_inf()) return p;
  Affine<F> q = q_in;
  if (neg) q.y = f_neg(q.y);
  if (p.is_inf()) return {q.x, q.y, F::one(), F::one()};
  auto U2 = q.x * p.ZZ;
  auto S2 = q.y * p.ZZZ;
  auto Pq = U2 - p.X;
  auto R = S2 - p.Y;
  if (f_eqz(Pq)) {
    if (f_eqz(R)) return xyzz_mdbl(q);
    return Xyzz<F>::inf();
  }
  auto PP = f_sqr(Pq);
  auto PPP = Pq * PP;
  auto Q = p.X * PP;
  auto X3 = f_sqr(R) - PPP - f_dbl(Q);
  auto Y3 = f_mulsub(R, Q - X3, p.Y, PPP);
  return {X3, Y3, p.ZZ * PP, p.ZZZ * PPP};
}

// add-2008-s: XYZZ + XYZZ, 12M + 2S.
template <class F>
NCG_DI Xyzz<F> xyzz_add(const Xyzz<F>& p, const Xyzz<F>& q) {
  if (q.is_inf()) return p;
  if (p.is_inf()) return q;
  auto U1 = p.X * q.ZZ;
  auto U2 = q.X * p.ZZ;
  auto S1 = p.Y * q.ZZZ;
  auto S2 = q.Y * p.ZZZ;
  auto Pq = U2 - U1;
  auto R = S2 - S1;
  if (f_eqz(Pq)) {
    if (f_eqz(R)) return xyzz_dbl(p);
    return Xyzz<F>::inf();
  }
  auto PP = f_sqr(Pq);
  auto PPP = Pq * PP;
  auto Q = U1 * PP;
  auto X3 = f_sqr(R) - PPP - f_dbl(Q);
  auto Y3 = f_mulsub(R, Q - X3, S1, PPP);
  return {X3, Y3, p.ZZ * q.ZZ * PP, p.ZZZ * q.ZZZ * PPP};
}

// ----------------------------------------------------------------- XYZZ over Fe9
template <class PR, int B>
NCG_DI Xyzz<Fe9<PR, B>> xyzz_mdbl(const Affine<Fe9<PR, B>>& p) {
  auto U = f_dbl(p.y);                                // bound 4
  auto V = f_sqr(U);
  auto W = U * V;
  auto S = p.x * V;
  auto xx = f_sqr(p.x);
  auto M = fe9_norm(xx + xx + xx);
  auto X3 = fe9_norm(f_sqr(M) - f_dbl(S));
  auto Y3 = M * (S - X3) - W * p.y;
  return {X3, Y3, V, W};
}
template <class PR, int B>
NCG_DI Xyzz<Fe9<PR, B>> xyzz_dbl(const Xyzz<Fe9<PR, B>>& p) {
  if (p.is_inf()) return p;
  auto U = f_dbl(p.Y);
  auto V = f_sqr(U);
  auto W = U * V;
  auto S = p.X * V;
  auto xx = f_sqr(p.X);
  auto M = fe9_norm(xx + xx + xx);
  auto X3 = fe9_norm(f_sqr(M) - f_dbl(S));
  auto Y3 = M * (S - X3) - W * p.Y;
  return {X3, Y3, V * p.ZZ, W * p.ZZZ};
}
template <class PR, int B>
NCG_DI Xyzz<Fe9<PR, B>> xyzz_madd(const Xyzz<Fe9<PR, B>>& p, const Affine<Fe9<PR, B>>& q_in, bool neg = false) {
  using F = Fe9<PR, B>;
  if (q_in.is_inf()) return p;
  const Fe9<PR, B + 1> qy = neg ? f_neg(q_in.y) : Fe9<PR, B + 1>(q_in.y);
  if (p.is_inf()) return {q_in.x, qy, F::one(), F::one()};
  auto U2 = q_in.x * p.ZZ;
  auto S2 = qy * p.ZZZ;
  auto Pw = U2 - p.X;
  auto Rw = S2 - p.Y;
  if (f_eqz(Pw)) {
    if (f_eqz(Rw)) return xyzz_mdbl(Affine<F>{q_in.x, qy});
    return Xyzz<F>::inf();
  }
  auto Pq = fe9_norm(Pw);
  auto R = fe9_norm(Rw);
  auto PP = f_sqr(Pq);
  auto PPP = Pq * PP;
  auto Q = p.X * PP;
  auto X3 = fe9_norm(f_sqr(R) - PPP - f_dbl(Q));
  auto Y3 = R * (Q - X3) - p.Y * PPP;
  return {X3, Y3, p.ZZ * PP, p.ZZZ * PPP};
}
template <class PR, int B>
NCG_DI Xyzz<Fe9<PR, B>> xyzz_add(const Xyzz<Fe9<PR, B>>& p, const Xyzz<Fe9<PR, B>>& q) {
  using F = Fe9<PR, B>;
  if (q.is_inf()) return p;
  if (p.is_inf()) return q;
  auto U1 = p.X * q.ZZ;
  auto U2 = q.X * p.ZZ;
  auto S1 = p.Y * q.ZZZ;
  auto S2 = q.Y * p.ZZZ;
  auto Pw = U2 - U1;
  auto Rw = S2 - S1;
  if (f_eqz(Pw)) {
    if (f_eqz(Rw)) return xyzz_dbl(p);
    return Xyzz<F>::inf();
  }
  auto Pq = fe9_norm(Pw);
  auto R = fe9_norm(Rw);
  auto PP = f_sqr(Pq);
  auto PPP = Pq * PP;
  auto Q = U1 * PP;
  auto X3 = fe9_norm(f_sqr(R) - PPP - f_dbl(Q));
  auto Y3 = R * (Q - X3) - S1 * PPP;
  return {X3, Y3, (p.ZZ * q.ZZ) * PP, (p.ZZZ * q.ZZZ) * PPP};
}

}  // namespace ncg
