// Curve traits for the kernels.  IDs match include/ncg.h.
#pragma once
#include "ec_sw.hpp"
#include "ec_te.hpp"
#include "endo.hpp"

namespace ncg {

enum CurveId : int { CURVE_SECP256K1 = 0, CURVE_ED25519 = 1, CURVE_BLS12_381_G1 = 2, CURVE_BLS12_381_G2 = 3 };

struct CurveSecp {  // src/secp256k1.ts:48-64
  using F = FeSecp;  // radix-2^29 lazy form (fe9.hpp)
  static constexpr bool GLV = true;  // h = 1: endomorphism split is always exact
  static constexpr int SCALAR_BITS = 256;
  static NCG_DI Fe9<Fe9SecpPR, 1> beta() { return Fe9<Fe9SecpPR, 1>::from_limbs(Fe9SecpPR::BETA); }
  static NCG_DI GlvSplit glv_split(const uint32_t (&k)[8]) { return secp_glv_split(k); }
};
struct CurveG1 {  // src/bls12-381.ts:134-148; no endomorphism in the reference (and inputs are
  using F = FeBls;  // not subgroup-checked), so none here either (SURVEY 8a gotcha 1)
  static constexpr bool GLV = false;
  static constexpr int SCALAR_BITS = 255;
  static NCG_DI F beta() { return F::one(); }
};
// CurveG1 for points KNOWN to lie in the prime-order subgroup (resident sets verified or decoded with the
// reference's isTorsionFree, endo.hpp): k = k1 + k2 z^2 and z^2 P = (beta x, -y), so the ladder's second stream
// runs on (beta x, y) with the sign of k2 flipped.  Never used for unverified inputs.
struct CurveG1E : CurveG1 {
  static constexpr bool GLV = true;
  static NCG_DI Fe29<1> beta() {
    Fe29<1> b;
#pragma unroll
    for (int i = 0; i < 14; i++) b.v[i] = ParamsBls29::G1_BETA[i];
    return b;
  }
  static NCG_DI GlvSplit glv_split(const uint32_t (&k)[8]) {
    uint32_t sub[2][6];
    bls_endo_split2(sub, k);
    GlvSplit s;
    s.k1neg = (sub[0][5] >> 31) != 0;
    const bool k2neg = (sub[1][5] >> 31) != 0;
    if (s.k1neg) mp_neg<6>(sub[0]);
    if (k2neg) mp_neg<6>(sub[1]);
#pragma unroll
    for (int i = 0; i < 5; i++) {
      s.k1[i] = sub[0][i];
      s.k2[i] = sub[1][i];
    }
    s.k2neg = !k2neg;  // the stream multiplies (beta x, y) = -(z^2 P)
    return s;
  }
};
struct CurveG2 {  // src/bls12-381.ts:321-345
  using F = FeBls2;
  static constexpr bool GLV = false;
  static constexpr int SCALAR_BITS = 255;
  static NCG_DI F beta() { return F::one(); }
};

// Device form of CurveG2 for the heavy kernels: one Fp2 element per lane PAIR (fe29.hpp), so
// kernels run 2 lanes per item (LANE_SHIFT).  Storage and wire formats are CurveG2's.
struct CurveG2P {
  using F = FeBls2P;
  static constexpr bool GLV = false;
  static constexpr int SCALAR_BITS = 255;
  static constexpr int LANE_SHIFT = 1;
  static NCG_DI F beta() { return F::one(); }
};
template <class C> struct LaneShift { static constexpr int value = 0; };
template <> struct LaneShift<CurveG2P> { static constexpr int value = 1; };
// curve type the device kernels are instantiated with
template <class C> struct DeviceCurve { using type = C; };
template <> struct DeviceCurve<CurveG2> { using type = CurveG2P; };

struct CurveEd {  // src/ed25519.ts:57-65 (twisted Edwards a = -1; cofactor 8: no subgroup tricks)
  using F = FEd;  // radix-2^29 lazy form (fe9.hpp)
  static constexpr bool GLV = false;
  static constexpr int SCALAR_BITS = 253;
};

// Affine wire point (canonical residues) -> Montgomery-form Affine<F>, and back.
template <class F>
NCG_DI Affine<F> load_affine_wire(const uint32_t* p) {
  constexpr int WW = FieldWire<F>::WORDS;
  Affine<F> a;
  a.x = FieldWire<F>::load(p);
  a.y = FieldWire<F>::load(p + WW);
  return a;
}
template <class F>
NCG_DI void store_affine_wire(uint32_t* p, const Affine<F>& a) {
  constexpr int WW = FieldWire<F>::WORDS;
  FieldWire<F>::store(p, a.x);
  FieldWire<F>::store(p + WW, a.y);
}

}  // namespace ncg
