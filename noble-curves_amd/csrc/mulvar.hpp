// Batch variable-base scalar multiplication: out[i] = k[i] * P[i], one lane per pair.
//
// Replaces, for a whole batch, the reference's Point.multiplyUnsafe(k)
// (src/abstract/weierstrass.ts:915-928 -> pushWnafPair :660-671 -> mulAddUnsafe
// src/abstract/curve.ts:820-836 -> wnafWalk :479-498) and - for the value - Point.multiply(k)
// (weierstrass.ts:900-907), whose result is the same group element, Z=1-normalised.
//
// Why not the reference's wNAF walk: on a 64-lane wavefront a data-dependent "add when the
// digit is non-zero" executes the add for the whole wave at nearly every bit.  The ladder here
// is regular instead: signed-odd fixed windows of W bits (every digit odd and non-zero), so all
// lanes do W doublings + one mixed add per scalar per window.  secp256k1 uses the GLV split
// k = k1 + lambda*k2 (two 128-bit halves sharing the doubling chain, psi(P) = (beta*x, y)) exactly
// as the reference does; bls12-381 G1/G2 get no endomorphism (the reference has none there and
// inputs may lie outside the prime-order subgroup).
//
// The per-lane table of odd multiples [1,3,..,2^W-1]*P holds *affine* points of an isomorphic
// curve (shared-Z "effective affine" trick), so every table add is a mixed add.  It lives in
// device memory (k_mul_var_gtab: item-major, one contiguous 1-3 KB block per lane - no LDS
// footprint, so wide windows and 3-4 waves/SIMD go together; the default) or in LDS (k_mul_var:
// limb-major, lane-minor, bank-conflict free for any digit pattern; used when no scratch is given).
#pragma once
#include "curves.hpp"
#include "scalar.hpp"

namespace ncg {

// One step of the table build T' = T + D (D = 2P) in co-Z form (Meloni's ZADDU): D and T share their Z, the sum comes out with
// Z' = Z (D.x - T.x) and D is RE-EXPRESSED at that Z' for free (W1, A1 are by-products), so the next step is co-Z again:
// 4M + 2S per table entry (a mixed Jacobian addition that also returns its Z ratio, the first form of this build, costs 8M + 3S:
// Z^2 and Z^3 products that co-Z operands do not need).  `zr` = D.x - T.x is the ratio Z' / Z; the running Z itself is not tracked
// here - the rescale pass of the table build multiplies the ratios up anyway.  With dx = D.x - T.x and dy = D.y - T.y:
// X3 = dy^2 - (D.x + T.x) dx^2, Y3 = dy (D.x dx^2 - X3) - D.y dx^3.
// The formula is incomplete: it is wrong when D = +-T (dx = 0), which happens while building [1,3,..]P exactly when P has small
// order (j*P = +-2P for some odd j < 2^W) - possible for any input the reference accepts on bls12-381 G1/G2 (cofactor > 1;
// Point.fromAffine does not subgroup-check, src/abstract/weierstrass.ts:710-718).  `degenerate` records that case; the caller
// then recomputes the lane with the complete ladder mul_var_slow below.
template <class F>
NCG_DI void coz_addu(Affine<F>& D, Affine<F>& T, F& zr, bool& degenerate) {
  auto dx = D.x - T.x;
  degenerate = degenerate || f_eqz(dx);
  auto C = f_sqr(dx);
  auto W1 = D.x * C;
  auto W2 = T.x * C;
  auto dy = D.y - T.y;
  auto A1 = D.y * (W1 - W2);
  auto X3 = f_sqr(dy) - W1 - W2;
  auto Y3 = dy * (W1 - X3) - A1;
  zr = dx * F::one();  // stored: bring the bound back under the storage bound
  D = {W1, A1};
  T = {X3, Y3};
}
template <class PR, int B>
NCG_DI void coz_addu(Affine<Fe9<PR, B>>& D, Affine<Fe9<PR, B>>& T, Fe9<PR, B>& zr, bool& degenerate) {
  auto dxw = D.x - T.x;
  degenerate = degenerate || f_eqz(dxw);
  auto dx = fe9_norm(dxw);
  auto dy = fe9_norm(D.y - T.y);
  auto C = f_sqr(dx);
  auto W1 = D.x * C;
  auto W2 = T.x * C;
  auto A1 = D.y * (W1 - W2);
  auto X3 = fe9_norm(f_sqr(dy) - W1 - W2);
  auto Y3 = dy * (W1 - X3) - A1;
  zr = dx;
  D = {W1, A1};
  T = {X3, Y3};
}

// Complete (every exceptional case handled by jac_madd / jac_dbl) MSB-first double-and-add over the
// whole 256-bit scalar: the value of the reference's multiplyUnsafe for ANY curve point
// (src/abstract/weierstrass.ts:915-928 on the complete formulas :793-880).  Only lanes whose
// window table degenerated (small-order P) come here, so its cost (256 dbl + ~128 madd) is
// irrelevant; P and k are re-read from memory so nothing stays live across the main ladder.
template <class C>
NCG_DI Jac<typename C::F> mul_var_slow(const uint32_t* __restrict__ pt_wire, const uint32_t* __restrict__ k_wire) {
  using F = typename C::F;
  const Affine<F> P = load_affine_wire<F>(pt_wire);
  Jac<F> R = Jac<F>::inf();
  for (int w = 7; w >= 0; w--) {
    const uint32_t word = k_wire[w];
    for (int bit = 31; bit >= 0; bit--) {
      R = jac_dbl(R);
      if ((word >> bit) & 1u) R = jac_madd(R, P);
    }
  }
  return R;
}

template <class C, int W>
struct MulVarCfg {
  using F = typename C::F;
  static constexpr int FW = FieldIO<F>::WORDS;    // stored words per field element (HBM scratch)
  static constexpr int TW = FieldIO<F>::LANE_WORDS;  // words per field element in this lane's LDS table
  static constexpr int WW = FieldWire<F>::WORDS;  // wire words per field element (HBM in/out)
  static constexpr int TS = 1 << (W - 1);                 // table entries: 1,3,..,2^W-1
  static constexpr int KBITS = C::GLV ? 129 : 257;        // bound on |k|+1 per stream
  static constexpr int M = (KBITS + W - 1) / W;           // windows
  static constexpr int NL = C::GLV ? 5 : 9;               // limbs of the window register
  static constexpr int LDS_WORDS = TS * 2 * TW * 64;      // per 64-lane block
};

// Per-lane body.  `tab` is this lane's table base, consecutive words `stride` apart
// (LDS: lds + lane, stride 64;  host unit test: a plain array, stride 1).
// JAC_OUT: write the Jacobian result (X, Y, Z in storage format, Z = 0 for infinity) to
// `out_wire` (3*FW words) and leave the inversion to k_jac_batch_affine; otherwise invert here
// and write the affine wire point.
// ZR_IN_TAB: keep the TS Z-ratios of the table build behind the table (TS more elements per lane)
// instead of in registers - for tables in device memory, where space is free and registers are not.
// PF (tables in device memory, inlined multiply, GLV curves; NCG_LADDER_PREFETCH): the table entry of the NEXT window travels
// into the wave's LDS slab (gfx950 global_load_lds_dwordx4, no register in between) while this window's doublings and
// additions run; `pf_slab` = 2 slots x 5 chunks x 64 lanes x 16 bytes of LDS, layout [slot][chunk][lane].
template <class C, int W, bool JAC_OUT = false, bool ZR_IN_TAB = false, bool PF = false, class TABPTR>
NCG_DI void mul_var_lane(const uint32_t* __restrict__ pt_wire, const uint32_t* __restrict__ k_wire,
                         uint32_t* __restrict__ out_wire, uint8_t* __restrict__ out_inf, bool active,
                         TABPTR tab, const int stride, uint32_t* pf_slab = nullptr) {
  using Cfg = MulVarCfg<C, W>;
  using F = typename C::F;
  constexpr int FW = Cfg::FW, TW = Cfg::TW, TS = Cfg::TS, M = Cfg::M, NL = Cfg::NL;

  Affine<F> P = load_affine_wire<F>(pt_wire);
  uint32_t k[8];
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = k_wire[i];
  const bool trivial_zero = P.is_inf() || mp_is_zero<8>(k);

  // ---- table of odd multiples on the isomorphic curve ------------------------------------
  // entry e, word w at tab[(e*2*FW + w)*stride]
  F Zg;
  bool degenerate = false;
  {
    Jac<F> D = jac_dbl(Jac<F>{P.x, P.y, F::one()});
    auto dz2 = f_sqr(D.Z);
    auto dz3 = dz2 * D.Z;
    Affine<F> Dc{D.X, D.Y};                 // 2P and P on the isomorphic curve, both at Z = 1: co-Z from the start
    Affine<F> T{P.x * dz2, P.y * dz3};
    F zr[ZR_IN_TAB ? 1 : TS];
    auto zr_put = [&](int j, const F& v) {
      if constexpr (ZR_IN_TAB) FieldIO<F>::store_strided(tab + (TS * 2 * TW + j * TW) * stride, stride, v);
      else zr[j] = v;
    };
    auto zr_get = [&](int j) -> F {
      if constexpr (ZR_IN_TAB) return FieldIO<F>::load_strided(tab + (TS * 2 * TW + j * TW) * stride, stride);
      else return zr[j];
    };
    FieldIO<F>::store_strided(tab, stride, T.x);
    FieldIO<F>::store_strided(tab + TW * stride, stride, T.y);
    // with the Z-ratios in memory nothing here indexes a register array: keep the loops rolled (the
    // unrolled build is 15 mixed additions of straight-line code in front of the ladder)
#pragma unroll(ZR_IN_TAB ? 1 : TS)
    for (int j = 1; j < TS; j++) {
      F zj;
      coz_addu(Dc, T, zj, degenerate);
      zr_put(j, zj);
      FieldIO<F>::store_strided(tab + (j * 2 * TW) * stride, stride, T.x);
      FieldIO<F>::store_strided(tab + (j * 2 * TW + TW) * stride, stride, T.y);
    }
    // bring every entry to the last entry's Z
    F s = F::one();
#pragma unroll(ZR_IN_TAB ? 1 : TS)
    for (int j = TS - 2; j >= 0; j--) {
      if (j == TS - 2) s = zr_get(j + 1);
      else s = s * zr_get(j + 1);
      auto s2 = f_sqr(s);
      auto s3 = s2 * s;
      F x = FieldIO<F>::load_strided(tab + (j * 2 * TW) * stride, stride);
      F y = FieldIO<F>::load_strided(tab + (j * 2 * TW + TW) * stride, stride);
      FieldIO<F>::store_strided(tab + (j * 2 * TW) * stride, stride, x * s2);
      FieldIO<F>::store_strided(tab + (j * 2 * TW + TW) * stride, stride, y * s3);
    }
    Zg = D.Z * s;   // s = the product of all the ratios = the Z the last entry was built at
  }

  // ---- scalar recoding ---------------------------------------------------------------------
  SignedOddWindows<NL, W, M> w1, w2;
  bool neg1 = false, neg2 = false;
  if constexpr (C::GLV) {
    GlvSplit gs = C::glv_split(k);
    w1.template init<5>(gs.k1);
    w2.template init<5>(gs.k2);
    neg1 = gs.k1neg;
    neg2 = gs.k2neg;
  } else {
    w1.template init<8>(k);
  }
  const auto beta = C::beta();

  // ---- ladder -------------------------------------------------------------------------------
  Jac<F> R = Jac<F>::inf();
#ifdef __HIP_DEVICE_COMPILE__
  if constexpr (PF && NCG_MUL_INLINE && C::GLV) {
    const int ln = threadIdx.x & 63;
    // entry `idx` of this lane's table -> slot `slot` (80 bytes: the 72 of (x, y) and 8 of whatever follows - another entry or
    // the Z-ratio area, both inside the lane's block)
    auto issue = [&](int slot, int idx) {
      const uint32_t* src = tab + idx * 2 * TW;
#pragma unroll
      for (int j = 0; j < 5; j++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * j),
                                         (__attribute__((address_space(3))) void*)(pf_slab + (slot * 5 + j) * 256), 16, 0, 0);
    };
    // the entry in `slot` has landed (every load of this wave was issued at least one mixed addition ago); once the reads
    // have returned the slot is free again
    auto fetch = [&](int slot, F& qx, F& qy) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      uint32_t wv[20];
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const uint4 v = *reinterpret_cast<const uint4*>(pf_slab + (slot * 5 + j) * 256 + ln * 4);
        wv[4 * j] = v.x; wv[4 * j + 1] = v.y; wv[4 * j + 2] = v.z; wv[4 * j + 3] = v.w;
      }
#pragma unroll
      for (int l = 0; l < 9; l++) {
        qx.v[l] = wv[l];
        qy.v[l] = wv[9 + l];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto entry_of = [](int d) { return ((d < 0 ? -d : d) - 1) >> 1; };
    int d1 = w1.pop(), d2 = w2.pop();
    issue(0, entry_of(d1));
    issue(1, entry_of(d2));
    for (int i = 0; i < M; i++) {
      if (i > 0) {
#pragma unroll 1
        for (int d = 0; d < W; d++) R = jac_dbl(R);
      }
#pragma unroll 1
      for (int e = 0; e < 2; e++) {
        const int d = e == 0 ? d1 : d2;
        const bool ng = e == 0 ? neg1 : neg2;
        F qx, qy;
        fetch(e, qx, qy);
        // refill the slot: the next digit of this stream, or - after the last window - entry 0 for the parity fix-up below
        int dn = 1;
        if (i + 1 < M) dn = e == 0 ? w1.pop() : w2.pop();
        issue(e, entry_of(dn));
        if (e == 0) d1 = dn; else d2 = dn;
        if (e == 1) qx = qx * beta;
        R = jac_madd_q(R, qx, f_cneg(qy, (d < 0) != ng));
      }
    }
    {
      F qx, qy;
      fetch(0, qx, qy);   // entry 0 (both slots hold it now; vmcnt(0) covers slot 1's load too)
      if (w1.was_even) R = jac_madd_q(R, qx, f_cneg(qy, !neg1));
      if (w2.was_even) R = jac_madd_q(R, qx * beta, f_cneg(qy, !neg2));
    }
  } else
#endif
  {
  for (int i = 0; i < M; i++) {
    if (i > 0) {
#pragma unroll(NCG_MUL_INLINE ? 1 : W)
      for (int d = 0; d < W; d++) R = jac_dbl(R);
    }
    // one mixed addition per stream
    if constexpr (NCG_MUL_INLINE && C::GLV) {
      // field multiply inlined: the two additions of the GLV pair are ONE loop body run twice (the window body - 4
      // doublings through one rolled body + this - then fits the 64 KB instruction cache; as straight-line code it
      // was 69 KB and re-fetched every window)
#pragma unroll 1
      for (int e = 0; e < 2; e++) {
        const int d = e == 0 ? w1.pop() : w2.pop();
        const bool ng = e == 0 ? neg1 : neg2;
        const int idx = ((d < 0 ? -d : d) - 1) >> 1;
        F qx = FieldIO<F>::load_strided(tab + (idx * 2 * TW) * stride, stride);
        const F qy = FieldIO<F>::load_strided(tab + (idx * 2 * TW + TW) * stride, stride);
        if (e == 1) qx = qx * beta;
        R = jac_madd_q(R, qx, f_cneg(qy, (d < 0) != ng));
      }
    } else {
      {
        int d1 = w1.pop();
        int e = ((d1 < 0 ? -d1 : d1) - 1) >> 1;
        const F qx = FieldIO<F>::load_strided(tab + (e * 2 * TW) * stride, stride);
        const F qy = FieldIO<F>::load_strided(tab + (e * 2 * TW + TW) * stride, stride);
        R = jac_madd_q(R, qx, f_cneg(qy, (d1 < 0) != neg1));
      }
      if constexpr (C::GLV) {
        int d2 = w2.pop();
        int e = ((d2 < 0 ? -d2 : d2) - 1) >> 1;
        const auto qx = FieldIO<F>::load_strided(tab + (e * 2 * TW) * stride, stride) * beta;
        const F qy = FieldIO<F>::load_strided(tab + (e * 2 * TW + TW) * stride, stride);
        R = jac_madd_q(R, qx, f_cneg(qy, (d2 < 0) != neg2));
      }
    }
  }
  // even scalars were bumped by one: take the extra point back out
  {
    const F qx = FieldIO<F>::load_strided(tab, stride);
    const F qy = FieldIO<F>::load_strided(tab + TW * stride, stride);
    if (w1.was_even) R = jac_madd_q(R, qx, f_cneg(qy, !neg1));
    if constexpr (C::GLV) {
      if (w2.was_even) R = jac_madd_q(R, qx * beta, f_cneg(qy, !neg2));
    }
  }
  }
  // back from the isomorphic curve, then to affine (weierstrass.ts:951-969 toAffine)
  const bool ladder_inf = R.is_inf();  // tested before the product: not every field keeps 0 * Zg literal
  R.Z = R.Z * Zg;
  if (ladder_inf) R = Jac<F>::inf();
  if (degenerate) R = mul_var_slow<C>(pt_wire, k_wire);  // small-order P: complete ladder instead
  bool inf = trivial_zero || R.is_inf();
  if constexpr (JAC_OUT) {
    if (inf) R = Jac<F>::inf();
    if (active) {
      FieldIO<F>::store(out_wire, R.X);
      FieldIO<F>::store(out_wire + FW, R.Y);
      FieldIO<F>::store(out_wire + 2 * FW, R.Z);
    }
    return;
  }
  Affine<F> A = jac_to_affine(R, f_inv(R.Z));
  if (inf) A = {F::zero(), F::zero()};
  if (active) {
    store_affine_wire<F>(out_wire, A);
    *out_inf = inf ? 1 : 0;
  }
}

// Batched Jacobian -> affine (the reference's normalizeZ / FpInvertBatch, src/abstract/curve.ts:
// 311-326, src/abstract/modular.ts:728-760): each lane runs Montgomery's trick over K
// consecutive points - one field inversion per K points instead of one per point.
// PRE_LDS: the K prefix products live in LDS (64-thread blocks, limb-major, lane-minor) instead of a per-lane array.
// In the translation units that inline the field multiply the array is not promoted to registers and ends up in
// scratch memory (592 bytes per lane for secp256k1, K = 16) - global-memory round trips in a dependent chain, which cost
// this kernel 0.22 ms per 2^20 on most boxes and 0.42 ms on the boxes of the pool with a slow memory path.
template <class C, int K, bool PRE_LDS = false>
__global__ void __launch_bounds__(PRE_LDS ? 64 : 256) k_jac_batch_affine(const uint32_t* __restrict__ jac,
                                                                         uint32_t* __restrict__ out_wire,
                                                                         uint8_t* __restrict__ out_inf, int n) {
  using F = typename C::F;
  constexpr int FW = FieldIO<F>::WORDS, WW = FieldWire<F>::WORDS, TW = FieldIO<F>::LANE_WORDS;
  extern __shared__ __attribute__((aligned(16))) uint32_t pre_lds[];
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> LaneShift<C>::value;
  const int i0 = t * K;
  if (i0 >= n) return;
  F pre[PRE_LDS ? 1 : K];   // pre[j] = product of the non-zero Z of points i0..i0+j-1
  auto pre_put = [&](int j, const F& v) {
    if constexpr (PRE_LDS) FieldIO<F>::store_strided(pre_lds + threadIdx.x + (size_t)j * TW * 64, 64, v);
    else pre[j] = v;
  };
  auto pre_get = [&](int j) -> F {
    if constexpr (PRE_LDS) return FieldIO<F>::load_strided(pre_lds + threadIdx.x + (size_t)j * TW * 64, 64);
    else return pre[j];
  };
  F acc = F::one();
#pragma unroll
  for (int j = 0; j < K; j++) {
    pre_put(j, acc);
    if (i0 + j < n) {
      F z = FieldIO<F>::load(jac + ((size_t)(i0 + j) * 3 + 2) * FW);
      if (!z.is_zero()) acc = acc * z;
    }
  }
  F inv = f_inv(acc);
#pragma unroll
  for (int j = K - 1; j >= 0; j--) {
    if (i0 + j < n) {
      const uint32_t* p = jac + (size_t)(i0 + j) * 3 * FW;
      F z = FieldIO<F>::load(p + 2 * FW);
      const bool inf = z.is_zero();
      Affine<F> A{F::zero(), F::zero()};
      if (!inf) {
        F zi = inv * pre_get(j);
        inv = inv * z;
        auto zi2 = f_sqr(zi);
        A = {FieldIO<F>::load(p) * zi2, FieldIO<F>::load(p + FW) * zi2 * zi};
      }
      store_affine_wire<F>(out_wire + (size_t)(i0 + j) * 2 * WW, A);
      out_inf[i0 + j] = inf ? 1 : 0;
    }
  }
}

// Batch normalisation of the reference's projective points (X, Y, Z) with x = X/Z, y = Y/Z -
// normalizeZ / toAffine(invZ) (src/abstract/curve.ts:311-326, src/abstract/weierstrass.ts:951-969,
// src/abstract/edwards.ts:595-609; the same map on Edwards (X, Y, Z, T)).  Wire input: X || Y || Z
// canonical residues; Z = 0 is the Weierstrass identity and comes out as (0, 0) with the flag
// set.  One inversion per K points (FpInvertBatch, src/abstract/modular.ts:728-760).
template <class F, int K>
__global__ void __launch_bounds__(256) k_proj_batch_affine(const uint32_t* __restrict__ proj_wire,
                                                           uint32_t* __restrict__ out_wire,
                                                           uint8_t* __restrict__ out_inf, int n) {
  constexpr int WW = FieldWire<F>::WORDS;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i0 = t * K;
  if (i0 >= n) return;
  F pre[K];
  F acc = F::one();
#pragma unroll
  for (int j = 0; j < K; j++) {
    pre[j] = acc;
    if (i0 + j < n) {
      F z = FieldWire<F>::load(proj_wire + ((size_t)(i0 + j) * 3 + 2) * WW);
      if (!f_eqz(z)) acc = acc * z;
    }
  }
  F inv = f_inv(acc);
#pragma unroll
  for (int j = K - 1; j >= 0; j--) {
    if (i0 + j < n) {
      const uint32_t* p = proj_wire + (size_t)(i0 + j) * 3 * WW;
      F z = FieldWire<F>::load(p + 2 * WW);
      const bool inf = f_eqz(z);
      F x = F::zero(), y = F::zero();
      if (!inf) {
        F zi = inv * pre[j];
        inv = inv * z;
        x = FieldWire<F>::load(p) * zi;
        y = FieldWire<F>::load(p + WW) * zi;
      }
      FieldWire<F>::store(out_wire + (size_t)(i0 + j) * 2 * WW, x);
      FieldWire<F>::store(out_wire + (size_t)(i0 + j) * 2 * WW + WW, y);
      out_inf[i0 + j] = inf ? 1 : 0;
    }
  }
}

// 1: the ladder of the GLV curves with the inlined multiply (the shipped secp256k1 kernel) prefetches its table entries through LDS
#ifndef NCG_LADDER_PREFETCH
#define NCG_LADDER_PREFETCH 0
#endif
template <class C> struct LadderPrefetch { static constexpr bool value = NCG_LADDER_PREFETCH != 0 && NCG_MUL_INLINE && C::GLV; };

// Variant with the per-lane table in device memory (item-major, each entry contiguous) instead of
// LDS: no LDS footprint, so the window width is no longer tied to occupancy.
template <class C, int W, int MINW, bool JAC_OUT>
__global__ void __launch_bounds__(64, MINW)
k_mul_var_gtab(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ scalars, uint32_t* __restrict__ out,
               uint8_t* __restrict__ out_inf, uint32_t* __restrict__ gtab, int n) {
  using Cfg = MulVarCfg<C, W>;
  constexpr int WW = Cfg::WW;
  constexpr int OUTW = JAC_OUT ? 3 * Cfg::FW : 2 * WW;
  const int lane_idx = blockIdx.x * 64 + threadIdx.x;  // table slot: one per LANE (a lane pair has two)
  const int idx = lane_idx >> LaneShift<C>::value;
  const bool active = idx < n;
  const int src = active ? idx : n - 1;
  constexpr bool PF = LadderPrefetch<C>::value;
  __shared__ __attribute__((aligned(16))) uint32_t pf_lds[PF ? 2 * 5 * 256 : 4];   // one wave per block
  mul_var_lane<C, W, JAC_OUT, true, PF>(pts + (size_t)src * 2 * WW, scalars + (size_t)src * 8, out + (size_t)src * OUTW,
                                        out_inf + src, active, gtab + (size_t)lane_idx * (Cfg::TS * 3 * Cfg::TW), 1, pf_lds);
}

template <class C, int W, int MINW, bool JAC_OUT>
__global__ void __launch_bounds__(64, MINW)
k_mul_var(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ scalars,
          uint32_t* __restrict__ out, uint8_t* __restrict__ out_inf, int n) {
  constexpr int WW = MulVarCfg<C, W>::WW;
  constexpr int OUTW = JAC_OUT ? 3 * MulVarCfg<C, W>::FW : 2 * WW;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x;
  const int idx = (blockIdx.x * 64 + lane) >> LaneShift<C>::value;
  const bool active = idx < n;
  const int src = active ? idx : n - 1;  // idle lanes redo the last item, stores masked
  mul_var_lane<C, W, JAC_OUT>(pts + (size_t)src * 2 * WW, scalars + (size_t)src * 8, out + (size_t)src * OUTW,
                              out_inf + src, active, lds + lane, 64);
}

// ---- launchers shared by the translation units that instantiate these kernels (host side)
// Table in device memory (k_mul_var_gtab): the table lives behind the Jacobian scratch in `jac_tmp`
// (mul_var_tmp_bytes accounts for both).
template <class C, int W>
inline size_t gtab_words_per_item() { return (size_t)MulVarCfg<C, W>::TS * 3 * MulVarCfg<C, W>::TW; }
inline size_t pad64(size_t n) { return (n + 63) / 64 * 64; }

template <class C, int W, int MINW, int K = 8>
inline hipError_t launch_mul_var_gtab(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf,
                                      int n, uint32_t* jac_tmp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  using Cfg = MulVarCfg<C, W>;
  constexpr int LS = LaneShift<C>::value;
  uint32_t* gtab = jac_tmp + pad64(n) * 3 * Cfg::FW;
  const unsigned blocks = (unsigned)((((size_t)n << LS) + 63) / 64);
  hipLaunchKernelGGL((k_mul_var_gtab<C, W, MINW, true>), dim3(blocks), dim3(64), 0, st, pts, scalars, jac_tmp, out_inf,
                     gtab, n);
  int threads = ((n + K - 1) / K) << LS;
  constexpr size_t pre_bytes = (size_t)K * FieldIO<typename C::F>::LANE_WORDS * 64 * 4;   // 36 KB for secp256k1, K = 16
  hipLaunchKernelGGL((k_jac_batch_affine<C, K, true>), dim3((threads + 63) / 64), dim3(64), pre_bytes, st, jac_tmp, out, out_inf, n);
  return hipGetLastError();
}

}  // namespace ncg
