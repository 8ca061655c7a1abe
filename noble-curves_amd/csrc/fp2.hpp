// Fp2 = Fp[u]/(u^2+1) over the bls12-381 base field, for G2.
// Reproduces the values of the reference's `_Field2` ops (src/abstract/tower.ts:393-438):
// mul = 3 Fp.mul Karatsuba (:420-431), sqr = 2 Fp.mul (:432-438).
#pragma once
#include "fe29.hpp"
#include "fe9.hpp"
#include "fp.hpp"

namespace ncg {

template <class PR>
struct Fp2T {
  using B = Fp<PR>;
  B c0, c1;
  static NCG_DI Fp2T zero() { return {B::zero(), B::zero()}; }
  static NCG_DI Fp2T one() { return {B::one(), B::zero()}; }
  NCG_DI bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  NCG_DI bool operator==(const Fp2T& o) const { return c0 == o.c0 && c1 == o.c1; }
  NCG_DI bool operator!=(const Fp2T& o) const { return !(*this == o); }
};

template <class PR>
NCG_DI Fp2T<PR> operator+(const Fp2T<PR>& a, const Fp2T<PR>& b) {  // tower.ts:404
  return {a.c0 + b.c0, a.c1 + b.c1};
}
template <class PR>
NCG_DI Fp2T<PR> operator-(const Fp2T<PR>& a, const Fp2T<PR>& b) {  // tower.ts:413
  return {a.c0 - b.c0, a.c1 - b.c1};
}
template <class PR>
NCG_DI Fp2T<PR> operator*(const Fp2T<PR>& a, const Fp2T<PR>& b) {  // tower.ts:420-431
  Fp<PR> t1 = a.c0 * b.c0;
  Fp<PR> t2 = a.c1 * b.c1;
  Fp<PR> m = (a.c0 + a.c1) * (b.c0 + b.c1);
  return {t1 - t2, m - (t1 + t2)};
}

using Fp2Bls = Fp2T<ParamsBlsP>;

// ---- uniform "field concept" used by the curve templates (works for Fp<PR> and Fp2T<PR>)
template <class PR> NCG_DI Fp<PR> f_sqr(const Fp<PR>& a) { return fp_sqr<PR>(a); }
template <class PR> NCG_DI Fp<PR> f_neg(const Fp<PR>& a) { return fp_neg<PR>(a); }
template <class PR> NCG_DI Fp<PR> f_dbl(const Fp<PR>& a) { return a + a; }
template <class PR> NCG_DI Fp<PR> f_inv(const Fp<PR>& a) { return fp_inv<PR>(a); }
template <class PR> NCG_DI Fp<PR> f_to_mont(const Fp<PR>& a) { return fp_to_mont<PR>(a); }
template <class PR> NCG_DI Fp<PR> f_from_mont(const Fp<PR>& a) { return fp_from_mont<PR>(a); }

template <class PR>
NCG_DI Fp2T<PR> f_sqr(const Fp2T<PR>& a) {  // tower.ts:432-438
  Fp<PR> s = a.c0 + a.c1;
  Fp<PR> d = a.c0 - a.c1;
  Fp<PR> c = a.c0 + a.c0;
  return {s * d, c * a.c1};
}
template <class PR> NCG_DI Fp2T<PR> f_neg(const Fp2T<PR>& a) { return {fp_neg<PR>(a.c0), fp_neg<PR>(a.c1)}; }
template <class PR> NCG_DI Fp2T<PR> f_dbl(const Fp2T<PR>& a) { return {a.c0 + a.c0, a.c1 + a.c1}; }
template <class PR>
NCG_DI Fp2T<PR> f_inv(const Fp2T<PR>& a) {  // tower.ts:458-475: (a - bu)/(a^2 + b^2)
  Fp<PR> f = fp_inv<PR>(fp_sqr<PR>(a.c0) + fp_sqr<PR>(a.c1));
  return {f * a.c0, f * fp_neg<PR>(a.c1)};
}
template <class PR> NCG_DI Fp2T<PR> f_to_mont(const Fp2T<PR>& a) { return {fp_to_mont<PR>(a.c0), fp_to_mont<PR>(a.c1)}; }
template <class PR> NCG_DI Fp2T<PR> f_from_mont(const Fp2T<PR>& a) { return {fp_from_mont<PR>(a.c0), fp_from_mont<PR>(a.c1)}; }

template <class PR> NCG_DI bool f_eqz(const Fp<PR>& a) { return a.is_zero(); }  // canonical residues
template <class PR> NCG_DI bool f_eqz(const Fp2T<PR>& a) { return a.is_zero(); }

// FieldIO<F>:   internal storage format (what kernels keep in HBM / LDS), WORDS 32-bit words
// FieldWire<F>: wire format of include/ncg.h (canonical residues, LE 32-bit limbs; Fp2 = c0, c1)
template <class F> struct FieldIO;
template <class PR>
struct FieldIO<Fp<PR>> {
  static constexpr int WORDS = PR::N;
  static constexpr int LANE_WORDS = WORDS;  // words one lane holds (differs for lane-paired Fp2)
  static NCG_DI Fp<PR> load(const uint32_t* p) { return fp_load<PR>(p); }
  static NCG_DI void store(uint32_t* p, const Fp<PR>& a) { fp_store<PR>(p, a); }
  // strided (word i at p[i*stride]) - LDS tables laid out limb-major, lane-minor
  template <class PTR> static NCG_DI Fp<PR> load_strided(PTR p, int stride) {
    Fp<PR> r;
#pragma unroll
    for (int i = 0; i < PR::N; i++) r.v[i] = p[i * stride];
    return r;
  }
  template <class PTR> static NCG_DI void store_strided(PTR p, int stride, const Fp<PR>& a) {
#pragma unroll
    for (int i = 0; i < PR::N; i++) p[i * stride] = a.v[i];
  }
};
template <class PR>
struct FieldIO<Fp2T<PR>> {
  static constexpr int WORDS = 2 * PR::N;
  static constexpr int LANE_WORDS = WORDS;
  static NCG_DI Fp2T<PR> load(const uint32_t* p) { return {fp_load<PR>(p), fp_load<PR>(p + PR::N)}; }
  static NCG_DI void store(uint32_t* p, const Fp2T<PR>& a) {
    fp_store<PR>(p, a.c0);
    fp_store<PR>(p + PR::N, a.c1);
  }
  template <class PTR> static NCG_DI Fp2T<PR> load_strided(PTR p, int stride) {
    return {FieldIO<Fp<PR>>::load_strided(p, stride), FieldIO<Fp<PR>>::load_strided(p + PR::N * stride, stride)};
  }
  template <class PTR> static NCG_DI void store_strided(PTR p, int stride, const Fp2T<PR>& a) {
    FieldIO<Fp<PR>>::store_strided(p, stride, a.c0);
    FieldIO<Fp<PR>>::store_strided(p + PR::N * stride, stride, a.c1);
  }
};

template <class F> struct FieldWire;
template <class PR>
struct FieldWire<Fp<PR>> {
  static constexpr int WORDS = PR::N;
  static NCG_DI Fp<PR> load(const uint32_t* p) { return fp_to_mont<PR>(fp_load<PR>(p)); }
  static NCG_DI void store(uint32_t* p, const Fp<PR>& a) { fp_store<PR>(p, fp_from_mont<PR>(a)); }
};
template <class PR>
struct FieldWire<Fp2T<PR>> {
  static constexpr int WORDS = 2 * PR::N;
  static NCG_DI Fp2T<PR> load(const uint32_t* p) {
    return {FieldWire<Fp<PR>>::load(p), FieldWire<Fp<PR>>::load(p + PR::N)};
  }
  static NCG_DI void store(uint32_t* p, const Fp2T<PR>& a) {
    FieldWire<Fp<PR>>::store(p, a.c0);
    FieldWire<Fp<PR>>::store(p + PR::N, a.c1);
  }
};

// ---- Fe9 adapters: 9 stored words (29-bit limbs), 8 wire words
template <class PR, int B>
struct FieldIO<Fe9<PR, B>> {
  static constexpr int WORDS = 9;
  static constexpr int LANE_WORDS = WORDS;
  static NCG_DI Fe9<PR, B> load(const uint32_t* p) {
    Fe9<PR, B> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = p[i];
    return r;
  }
  static NCG_DI void store(uint32_t* p, const Fe9<PR, B>& a) {
#pragma unroll
    for (int i = 0; i < 9; i++) p[i] = a.v[i];
  }
  template <class PTR> static NCG_DI Fe9<PR, B> load_strided(PTR p, int stride) {
    Fe9<PR, B> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = p[i * stride];
    return r;
  }
  template <class PTR> static NCG_DI void store_strided(PTR p, int stride, const Fe9<PR, B>& a) {
#pragma unroll
    for (int i = 0; i < 9; i++) p[i * stride] = a.v[i];
  }
};
template <class PR, int B>
struct FieldWire<Fe9<PR, B>> {
  static constexpr int WORDS = 8;
  static NCG_DI Fe9<PR, B> load(const uint32_t* p) { return fe9_from_wire<PR>(p); }
  static NCG_DI void store(uint32_t* p, const Fe9<PR, B>& a) { fe9_to_wire(p, a); }
};

// ---- Fe29 / Fe29x2 adapters
template <int B>
struct FieldIO<Fe29<B>> {
  static constexpr int WORDS = 14;
  static constexpr int LANE_WORDS = WORDS;
  static NCG_DI Fe29<B> load(const uint32_t* p) {
    Fe29<B> r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.v[i] = p[i];
    return r;
  }
  static NCG_DI void store(uint32_t* p, const Fe29<B>& a) {
#pragma unroll
    for (int i = 0; i < 14; i++) p[i] = a.v[i];
  }
  template <class PTR> static NCG_DI Fe29<B> load_strided(PTR p, int stride) {
    Fe29<B> r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.v[i] = p[i * stride];
    return r;
  }
  template <class PTR> static NCG_DI void store_strided(PTR p, int stride, const Fe29<B>& a) {
#pragma unroll
    for (int i = 0; i < 14; i++) p[i * stride] = a.v[i];
  }
};
template <int B>
struct FieldIO<Fe29x2<B>> {
  static constexpr int WORDS = 28;
  static constexpr int LANE_WORDS = WORDS;
  static NCG_DI Fe29x2<B> load(const uint32_t* p) { return {FieldIO<Fe29<B>>::load(p), FieldIO<Fe29<B>>::load(p + 14)}; }
  static NCG_DI void store(uint32_t* p, const Fe29x2<B>& a) {
    FieldIO<Fe29<B>>::store(p, a.c0);
    FieldIO<Fe29<B>>::store(p + 14, a.c1);
  }
  template <class PTR> static NCG_DI Fe29x2<B> load_strided(PTR p, int stride) {
    return {FieldIO<Fe29<B>>::load_strided(p, stride), FieldIO<Fe29<B>>::load_strided(p + 14 * stride, stride)};
  }
  template <class PTR> static NCG_DI void store_strided(PTR p, int stride, const Fe29x2<B>& a) {
    FieldIO<Fe29<B>>::store_strided(p, stride, a.c0);
    FieldIO<Fe29<B>>::store_strided(p + 14 * stride, stride, a.c1);
  }
};
// lane-paired Fp2: same storage as Fe29x2 (c0 then c1, 14 words each); each lane moves its half
template <int B>
struct FieldIO<Fe29x2P<B>> {
  static constexpr int WORDS = 28;
  static constexpr int LANE_WORDS = 14;  // per-lane tables (LDS) hold this lane's half only
  template <class PTR> static NCG_DI Fe29x2P<B> load_strided(PTR p, int stride) {
    return Fe29x2P<B>(FieldIO<Fe29<B>>::load_strided(p, stride));
  }
  template <class PTR> static NCG_DI void store_strided(PTR p, int stride, const Fe29x2P<B>& a) {
    FieldIO<Fe29<B>>::store_strided(p, stride, a.h);
  }
  static NCG_DI Fe29x2P<B> load(const uint32_t* p) { return Fe29x2P<B>(FieldIO<Fe29<B>>::load(p + (pair_odd() ? 14 : 0))); }
  static NCG_DI void store(uint32_t* p, const Fe29x2P<B>& a) { FieldIO<Fe29<B>>::store(p + (pair_odd() ? 14 : 0), a.h); }
};
template <int B>
struct FieldWire<Fe29x2P<B>> {
  static constexpr int WORDS = 24;
  static NCG_DI Fe29x2P<B> load(const uint32_t* p) { return Fe29x2P<B>(Fe29<B>(fe29_from_wire(p + (pair_odd() ? 12 : 0)))); }
  static NCG_DI void store(uint32_t* p, const Fe29x2P<B>& a) { fe29_to_wire(p + (pair_odd() ? 12 : 0), a.h); }
};
template <int B>
struct FieldWire<Fe29<B>> {
  static constexpr int WORDS = 12;
  static NCG_DI Fe29<B> load(const uint32_t* p) { return fe29_from_wire(p); }
  static NCG_DI void store(uint32_t* p, const Fe29<B>& a) { fe29_to_wire(p, a); }
};
template <int B>
struct FieldWire<Fe29x2<B>> {
  static constexpr int WORDS = 24;
  static NCG_DI Fe29x2<B> load(const uint32_t* p) {
    Fe29<B> a = fe29_from_wire(p), b = fe29_from_wire(p + 12);
    return {a, b};
  }
  static NCG_DI void store(uint32_t* p, const Fe29x2<B>& a) {
    fe29_to_wire(p, a.c0);
    fe29_to_wire(p + 12, a.c1);
  }
};

// conditional negation; the result type is f_neg's (lazy fields widen their bound there)
template <class T>
NCG_DI auto f_cneg(const T& a, bool c) -> decltype(f_neg(a)) {
  using R = decltype(f_neg(a));
  return c ? f_neg(a) : R(a);
}

}  // namespace ncg
