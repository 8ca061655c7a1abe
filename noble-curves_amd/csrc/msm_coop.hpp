// Latency-optimised XYZZ addition / doubling for the narrow end of the MSM (late fold levels, the
// grouping of the pending sums, the multi-GPU combine): FOUR work items - lanes, or lane pairs of
// the paired Fp2 form - share ONE group operation.
//
// Why: the fold of the bucket sums (curve.ts:895-900 restated as a log-depth tree) ends in levels
// that hold fewer additions than the chip has SIMDs, so each level costs the latency of one
// dependent addition on a lone wave: 14 field multiplications back to back, 13-16 us on gfx950
// (a lone wave already issues its multiply-adds at the per-wave rate, so neither occupancy nor
// instruction-level parallelism shortens it).  The 12M + 2S of add-2008-s have dependency depth
// four, though:
//     step 1   U1 = X1 ZZ2     U2 = X2 ZZ1     S1 = Y1 ZZZ2     S2 = Y2 ZZZ1
//     step 2   PP = P^2        RR = R^2        ZZ12 = ZZ1 ZZ2   ZZZ12 = ZZZ1 ZZZ2     (P = U2 - U1, R = S2 - S1)
//     step 3   PPP = P PP      Q = U1 PP       ZZ3 = ZZ12 PP    -
//     step 4   A = R (Q - X3)  B = S1 PPP      -                ZZZ3 = ZZZ12 PPP      (X3 = RR - PPP - 2 Q)
//     out      X3              Y3 = A - B      ZZ3              ZZZ3
// One column per item ("role" 0..3); the items exchange products through an LDS scratch of 10 field
// elements per group.  Every lane executes the same instruction stream - the roles only select
// operands (v_cndmask) - so there is no divergence inside a step; four multiplication times instead
// of fourteen.  The exceptional cases of the group law (either operand the identity, P = +-Q) are
// detected group-uniformly and routed to a copy / to the complete single-lane routine.
// Doubling (dbl-2008-s-1): V = (2Y)^2, XX = X^2 | W = U V, S = X V, ZZ3 = V ZZ, MM = (3 XX)^2 |
// A = M (S - X3), B = W Y, ZZZ3 = W ZZZ: three multiplication times instead of nine.
//
// Only for the Fe29-based Weierstrass groups (bls12-381 G1 and the lane-paired G2), whose lazy value
// bounds make `a - b` safe without normalisation; secp256k1 / ed25519 keep the single-lane routines.
#pragma once
#include <utility>

#include "msm.hpp"

namespace ncg {

template <class C> struct CoopOK { static constexpr bool value = false; };
template <> struct CoopOK<CurveG1> { static constexpr bool value = true; };
template <> struct CoopOK<CurveG2P> { static constexpr bool value = true; };

template <class F, int B> struct Fe29Bound;   // the same field family at another value bound
template <int A, int B> struct Fe29Bound<Fe29<A>, B> { using type = Fe29<B>; };
template <int A, int B> struct Fe29Bound<Fe29x2P<A>, B> { using type = Fe29x2P<B>; };

template <int B>
NCG_DI Fe29<B> coop_sel(bool c, const Fe29<B>& a, const Fe29<B>& b) { return fe29_select(c, a, b); }
template <int B>
NCG_DI Fe29x2P<B> coop_sel(bool c, const Fe29x2P<B>& a, const Fe29x2P<B>& b) { return Fe29x2P<B>(fe29_select(c, a.h, b.h)); }

constexpr int COOP_SLOTS = 10;

#ifdef __HIP_DEVICE_COMPILE__
// orders the LDS traffic of the items of one group; all of them sit in the same wave, whose DS
// operations execute in program order, so no hardware barrier is needed - only the compiler must
// not move a read above the write it depends on
__device__ __forceinline__ void coop_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the complete single-lane routine for the exceptional cases: out of line, so that its registers (and spills) stay
// out of the cooperative path
template <class C>
__device__ __noinline__ void coop_complete_add(const uint32_t* p, const uint32_t* q, uint32_t* out) {
  using G = MsmGroup<C>;
  G::acc_store(out, G::add(G::acc_load(p), G::acc_load(q)));
}

template <class C>
struct CoopXyzz {
  using G = MsmGroup<C>;
  using F = typename C::F;                                            // stored coordinates (value bound 64)
  using F2 = decltype(std::declval<F>() * std::declval<F>());        // a product (bound 2)
  using FD = decltype(std::declval<F>() - std::declval<F>());        // a difference of stored values
  static constexpr int LS = LaneShift<C>::value;
  static constexpr int FW = G::FW;
  static constexpr int GROUP_WORDS = COOP_SLOTS * FW;
  static constexpr int GROUP_LANES = 4 << LS;

  static __device__ __forceinline__ int role() { return (int)((threadIdx.x >> LS) & 3u); }
  static __device__ __forceinline__ int group_in_block() { return (int)(threadIdx.x >> (LS + 2)); }
  // lane (within the wave) of item `r` of my group
  static __device__ __forceinline__ int lane_of(int r) { return (int)((threadIdx.x & 63u & ~(unsigned)(GROUP_LANES - 1)) + ((unsigned)r << LS)); }
  static __device__ __forceinline__ bool bcast(bool v, int r) { return __shfl((int)v, lane_of(r)) != 0; }
  static __device__ __forceinline__ F ld(const uint32_t* lds, int slot) { return FieldIO<F>::load(lds + slot * FW); }
  template <class T> static __device__ __forceinline__ void st(uint32_t* lds, int slot, const T& v) { FieldIO<T>::store(lds + slot * FW, v); }

  // out = p + q.  All four items of the group call this with the same arguments; `lds` = the group's scratch.
  // out may alias p or q.
  static __device__ __forceinline__ void add(uint32_t* lds, const uint32_t* p, const uint32_t* q, uint32_t* out) {
    const int r = role();
    // ---- step 1
    const uint32_t* pa = r == 0 ? p : r == 1 ? q : r == 2 ? p + FW : q + FW;                      // X1 X2 Y1 Y2
    const uint32_t* pb = r == 0 ? q + 2 * FW : r == 1 ? p + 2 * FW : r == 2 ? q + 3 * FW : p + 3 * FW;  // ZZ2 ZZ1 ZZZ2 ZZZ1
    const F a = FieldIO<F>::load(pa), b = FieldIO<F>::load(pb);
    const bool bz = b.is_zero();
    const bool q_inf = bcast(bz, 0), p_inf = bcast(bz, 1);
    if (p_inf || q_inf) {  // group-uniform: the other operand, coordinate by coordinate
      const uint32_t* src = q_inf ? p : q;
      const F v = FieldIO<F>::load(src + r * FW);
      FieldIO<F>::store(out + r * FW, v);
      return;
    }
    const F2 r1 = a * b;
    st(lds, r, r1);       // 0 U1, 1 U2, 2 S1, 3 S2
    st(lds, 4 + r, b);    // 4 ZZ2, 5 ZZ1, 6 ZZZ2, 7 ZZZ1
    coop_sync();
    // ---- step 2
    const int i1 = r == 0 ? 1 : r == 1 ? 3 : r == 2 ? 5 : 7, i2 = i1 - 1;
    const F v1 = ld(lds, i1), v2 = ld(lds, i2);
    const FD d = v1 - v2;                                   // role 0: P, role 1: R
    const bool dz = f_eqz(d);
    const bool p_zero = bcast(dz, 0);
    if (p_zero) {  // P = +-Q: the complete routine on one item (doubling / identity)
      if (r == 0) coop_complete_add<C>(p, q, out);
      return;
    }
    const FD x2 = coop_sel(r < 2, d, FD(v1));
    const FD y2 = coop_sel(r < 2, d, FD(v2));
    const F2 r2 = x2 * y2;                                  // PP, RR, ZZ12, ZZZ12
    coop_sync();
    st(lds, 4 + r, r2);   // 4 PP, 5 RR, 6 ZZ12, 7 ZZZ12
    coop_sync();
    // ---- step 3
    const F u1 = ld(lds, 0), pp = ld(lds, 4);
    const FD x3 = coop_sel(r == 0, d, coop_sel(r == 1, FD(u1), FD(r2)));
    const F2 r3 = x3 * pp;                                  // PPP, Q, ZZ3, (unused)
    if (r < 2) st(lds, 8 + r, r3);  // 8 PPP, 9 Q
    coop_sync();
    // ---- step 4
    const F2 rr = FieldIO<F2>::load(lds + 5 * FW), ppp = FieldIO<F2>::load(lds + 8 * FW), qq = FieldIO<F2>::load(lds + 9 * FW);
    const F s1 = ld(lds, 2), s2 = ld(lds, 3);
    const auto X3 = rr - ppp - f_dbl(qq);                   // bound 8
    const auto qx = qq - X3;                                // bound 10
    using FY = Fe29Bound<F, 16>;
    const FD x4 = coop_sel(r == 0, s2 - s1, coop_sel(r == 1, FD(s1), FD(r2)));
    const typename FY::type y4 = coop_sel(r == 0, typename FY::type(qx), typename FY::type(ppp));
    const F2 r4 = x4 * y4;                                  // A, B, (unused), ZZZ3
    if (r == 0) st(lds, 4, r4);
    coop_sync();
    const F2 A = FieldIO<F2>::load(lds + 4 * FW);
    // ---- out
    if (r == 0) {
      FieldIO<F>::store(out, F(X3));
    } else if (r == 1) {
      FieldIO<F>::store(out + FW, F(A - r4));
    } else if (r == 2) {
      FieldIO<F>::store(out + 2 * FW, F(r3));
    } else {
      FieldIO<F>::store(out + 3 * FW, F(r4));
    }
  }

  // out = 2 p (out may alias p)
  static __device__ __forceinline__ void dbl(uint32_t* lds, const uint32_t* p, uint32_t* out) {
    const int r = role();
    const F zz = FieldIO<F>::load(p + 2 * FW);
    if (zz.is_zero()) {  // identity (same verdict on every item: they all read ZZ)
      const F v = FieldIO<F>::load(p + r * FW);
      FieldIO<F>::store(out + r * FW, v);
      return;
    }
    const F X = FieldIO<F>::load(p), Y = FieldIO<F>::load(p + FW), zzz = FieldIO<F>::load(p + 3 * FW);
    const auto U = f_dbl(Y);                                 // bound 128
    // ---- step 1: V = U^2 (role 0), XX = X^2 (role 1)
    using FU = decltype(U);
    const FU x1 = coop_sel(r == 0, U, FU(X));
    const F2 r1 = x1 * x1;
    if (r < 2) st(lds, r, r1);  // 0 V, 1 XX
    coop_sync();
    const F2 V = FieldIO<F2>::load(lds), XX = FieldIO<F2>::load(lds + FW);
    const auto M = f_dbl(XX) + XX;                           // bound 6
    // ---- step 2: W = U V, S = X V, ZZ3 = V ZZ, MM = M^2
    const FU x2 = coop_sel(r == 0, U, coop_sel(r == 1, FU(X), coop_sel(r == 2, FU(zz), FU(M))));
    using FM = decltype(M);
    const FM y2 = coop_sel(r == 3, M, FM(V));
    const F2 r2 = x2 * y2;
    coop_sync();
    st(lds, 2 + r, r2);         // 2 W, 3 S, 4 ZZ3, 5 MM
    coop_sync();
    const F2 W = FieldIO<F2>::load(lds + 2 * FW), S = FieldIO<F2>::load(lds + 3 * FW), MM = FieldIO<F2>::load(lds + 5 * FW);
    const auto X3 = MM - f_dbl(S);                           // bound 2 + 4 = 6
    const auto sx = S - X3;                                  // bound 2 + 8 = 10
    // ---- step 3: A = M (S - X3), B = W Y, ZZZ3 = W ZZZ
    using FS = Fe29Bound<F, 16>;
    const typename FS::type x3 = coop_sel(r == 0, typename FS::type(M), typename FS::type(W));
    using FB = Fe29Bound<F, 64>;
    const typename FB::type y3 = coop_sel(r == 0, typename FB::type(sx), coop_sel(r == 1, typename FB::type(Y), typename FB::type(zzz)));
    const F2 r3 = x3 * y3;      // A, B, ZZZ3 (role 2), unused
    if (r == 0) st(lds, 6, r3);
    coop_sync();
    const F2 A = FieldIO<F2>::load(lds + 6 * FW);
    if (r == 0) {
      FieldIO<F>::store(out, F(X3));
    } else if (r == 1) {
      FieldIO<F>::store(out + FW, F(A - r3));
    } else if (r == 2) {
      FieldIO<F>::store(out + 3 * FW, F(r3));
    } else {
      FieldIO<F>::store(out + 2 * FW, F(FieldIO<F2>::load(lds + 4 * FW)));
    }
  }

  // out = p (coordinate by coordinate)
  static __device__ __forceinline__ void copy(const uint32_t* p, uint32_t* out) {
    const int r = role();
    const F v = FieldIO<F>::load(p + r * FW);
    FieldIO<F>::store(out + r * FW, v);
  }
};
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace ncg
