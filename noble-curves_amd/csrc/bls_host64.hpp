// HOST side of the bls12-381 MSM: the serial Horner combine of the window sums (src/abstract/curve.ts:901-902:
// `sum = sum.add(resI); sum = sum.double() x c`) and the final toAffine (weierstrass.ts:951-969).
//
// 255 dependent doublings are a latency chain: one GPU lane needs ~10 us per doubling, a CPU core well under
// 0.2 us, so this stays on the host - and is written for the host: 6 x 64-bit limbs,
// Montgomery R = 2^384 (CIOS with 128-bit products), Jacobian coordinates (dbl-2009-l 2M + 5S, add-2007-bl
// 11M + 5S), values canonical in [0, p) so the exceptional cases of the group law are exact comparisons.
// The device hands over XYZZ accumulators in its own storage format (radix 2^29, R = 2^406, lazily reduced);
// `from_fe29` canonicalises and changes the Montgomery radix with one multiplication by 2^362.
// Round 2 ran this step through the device templates compiled for the host (radix 2^58 twin of the 29-bit
// form, XYZZ): 0.19 ms (G1) / 0.72 ms (G2) per MSM; this form: see DESIGN.md section 5.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "fe29.hpp"

namespace ncg {
namespace h64 {

typedef unsigned __int128 u128;

struct Fp {
  uint64_t v[6];
};

struct Consts {
  uint64_t p[6];
  uint64_t inv;   // -p^-1 mod 2^64
  uint64_t pinv_pos;  // p^-1 mod 2^64
  Fp one;         // 2^384 mod p
  Fp raw_one;     // the integer 1
};

inline bool geq(const uint64_t* a, const uint64_t* b) {
  for (int i = 5; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return true;
}
inline void sub_in_place(uint64_t* a, const uint64_t* b) {
  u128 br = 0;
  for (int i = 0; i < 6; i++) {
    u128 d = (u128)a[i] - b[i] - (uint64_t)br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
inline void dbl_mod(uint64_t* a, const uint64_t* p) {  // a = 2a mod p for a < p < 2^383
  uint64_t c = 0;
  for (int i = 0; i < 6; i++) {
    uint64_t n = (a[i] << 1) | c;
    c = a[i] >> 63;
    a[i] = n;
  }
  if (geq(a, p)) sub_in_place(a, p);
}

inline const Consts& K() {
  static const Consts k = [] {
    Consts c;
    for (int i = 0; i < 6; i++) c.p[i] = (uint64_t)ParamsBlsP::P[2 * i] | ((uint64_t)ParamsBlsP::P[2 * i + 1] << 32);
    uint64_t x = 1;  // Newton: x = p^-1 mod 2^64
    for (int i = 0; i < 6; i++) x *= 2 - c.p[0] * x;
    c.inv = 0 - x;
    c.pinv_pos = x;
    uint64_t t[6] = {1, 0, 0, 0, 0, 0};
    memset(&c.raw_one, 0, sizeof c.raw_one);
    c.raw_one.v[0] = 1;
    for (int i = 0; i < 384; i++) {
      dbl_mod(t, c.p);
    }
    memcpy(c.one.v, t, sizeof t);
    return c;
  }();
  return k;
}

inline bool is_zero(const Fp& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4] | a.v[5]) == 0; }
inline bool eq(const Fp& a, const Fp& b) {
  uint64_t d = 0;
  for (int i = 0; i < 6; i++) d |= a.v[i] ^ b.v[i];
  return d == 0;
}
inline Fp zero() {
  Fp r;
  memset(&r, 0, sizeof r);
  return r;
}

// Montgomery product a b 2^-384 mod p, result in [0, p).  CIOS with the multiplication and the reduction of a row fused
// into one pass over the limbs ("no-carry" form: the top word of p is below 2^63 - 1, so the running value never needs a
// seventh / eighth word): per row 12 64 x 64 -> 128 products and two carry chains that the compiler keeps in registers -
// 1.6x the speed of the two-pass form this replaced (tests/test_msm_finish_host.py pins both curves on the oracle).
inline Fp mul_c(const Fp& a, const Fp& b) {
  const Consts& k = K();
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
#define NCG_H64_ROW(bi)                                                        \
  {                                                                            \
    u128 A = (u128)a.v[0] * (bi) + t0;                                         \
    const uint64_t m = (uint64_t)A * k.inv;                                    \
    u128 C = (u128)m * k.p[0] + (uint64_t)A;                                   \
    A >>= 64;                                                                  \
    C >>= 64;                                                                  \
    A += (u128)a.v[1] * (bi) + t1;                                             \
    C += (u128)m * k.p[1] + (uint64_t)A;                                       \
    t0 = (uint64_t)C;                                                          \
    A >>= 64;                                                                  \
    C >>= 64;                                                                  \
    A += (u128)a.v[2] * (bi) + t2;                                             \
    C += (u128)m * k.p[2] + (uint64_t)A;                                       \
    t1 = (uint64_t)C;                                                          \
    A >>= 64;                                                                  \
    C >>= 64;                                                                  \
    A += (u128)a.v[3] * (bi) + t3;                                             \
    C += (u128)m * k.p[3] + (uint64_t)A;                                       \
    t2 = (uint64_t)C;                                                          \
    A >>= 64;                                                                  \
    C >>= 64;                                                                  \
    A += (u128)a.v[4] * (bi) + t4;                                             \
    C += (u128)m * k.p[4] + (uint64_t)A;                                       \
    t3 = (uint64_t)C;                                                          \
    A >>= 64;                                                                  \
    C >>= 64;                                                                  \
    A += (u128)a.v[5] * (bi) + t5;                                             \
    C += (u128)m * k.p[5] + (uint64_t)A;                                       \
    t4 = (uint64_t)C;                                                          \
    t5 = (uint64_t)(C >> 64) + (uint64_t)(A >> 64);                            \
  }
  NCG_H64_ROW(b.v[0])
  NCG_H64_ROW(b.v[1])
  NCG_H64_ROW(b.v[2])
  NCG_H64_ROW(b.v[3])
  NCG_H64_ROW(b.v[4])
  NCG_H64_ROW(b.v[5])
#undef NCG_H64_ROW
  const uint64_t t[6] = {t0, t1, t2, t3, t4, t5};
  unsigned long long d[6], bw = 0;
  for (int i = 0; i < 6; i++) d[i] = __builtin_subcll(t[i], k.p[i], bw, &bw);
  const uint64_t keep = 0 - (uint64_t)bw;   // borrow: t < p, keep t (t < 2p always)
  Fp r;
  for (int i = 0; i < 6; i++) r.v[i] = (t[i] & keep) | (d[i] & ~keep);
  return r;
}

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define NCG_H64_ADX 1
// The same product (same row structure: multiply row, then reduction row, "no-carry" form) on the two independent carry
// chains of BMI2 / ADX: MULX leaves the flags alone, ADCX carries through CF and ADOX through OF, so the high halves and
// the low halves of a row are added in one pass without saving a carry.  The compiler's code for mul_c keeps the two
// chains in one flag (setc / movzx per word): 24-28 ns per product on the GPU boxes' EPYC 9575F against 13-14 ns here.
// Chosen at run time (mul, below); hosts without the extensions run mul_c.  Values identical (tests/test_msm_finish_host.py
// runs both).
#define NCG_H64_MUL_STEP(j, off) \
  "adcxq %[A], %[t" #j "]\n\t"   \
  "mulxq " #off "(%[a]), %%rax, %[A]\n\t" \
  "adoxq %%rax, %[t" #j "]\n\t"
#define NCG_H64_MUL_ROW(boff)                   \
  "xorl %%eax, %%eax\n\t"                       \
  "movq " #boff "(%[b]), %%rdx\n\t"             \
  "mulxq 0(%[a]), %%rax, %[A]\n\t"              \
  "adoxq %%rax, %[t0]\n\t"                      \
  NCG_H64_MUL_STEP(1, 8) NCG_H64_MUL_STEP(2, 16) NCG_H64_MUL_STEP(3, 24) NCG_H64_MUL_STEP(4, 32) NCG_H64_MUL_STEP(5, 40) \
  "movl $0, %%eax\n\t"                          \
  "adcxq %%rax, %[A]\n\t"                       \
  "adoxq %%rax, %[A]\n\t"
#define NCG_H64_RED_STEP(j, jm1, off)           \
  "adcxq %[t" #j "], %[t" #jm1 "]\n\t"          \
  "mulxq " #off "(%[k]), %%rax, %[t" #j "]\n\t" \
  "adoxq %%rax, %[t" #jm1 "]\n\t"
#define NCG_H64_RED_ROW                          \
  "movq %[t0], %%rdx\n\t"                       \
  "imulq 48(%[k]), %%rdx\n\t"                   \
  "xorl %%eax, %%eax\n\t"                       \
  "mulxq 0(%[k]), %%rax, %[h]\n\t"              \
  "adcxq %[t0], %%rax\n\t"                      \
  "movq %[h], %[t0]\n\t"                        \
  NCG_H64_RED_STEP(1, 0, 8) NCG_H64_RED_STEP(2, 1, 16) NCG_H64_RED_STEP(3, 2, 24) NCG_H64_RED_STEP(4, 3, 32) NCG_H64_RED_STEP(5, 4, 40) \
  "movl $0, %%eax\n\t"                          \
  "adcxq %%rax, %[t5]\n\t"                      \
  "adoxq %[A], %[t5]\n\t"
__attribute__((target("bmi2,adx"), noinline)) inline Fp mul_adx(const Fp& a, const Fp& b) {
  const Consts& k = K();
  static_assert(offsetof(Consts, inv) == 48, "the asm addresses k.inv at 48(k)");
  uint64_t t0, t1, t2, t3, t4, t5, A, h;
  asm("xorl %%eax, %%eax\n\t"
      "movq 0(%[b]), %%rdx\n\t"
      "mulxq 0(%[a]), %[t0], %[t1]\n\t"
      "mulxq 8(%[a]), %%rax, %[t2]\n\t"
      "adcxq %%rax, %[t1]\n\t"
      "mulxq 16(%[a]), %%rax, %[t3]\n\t"
      "adcxq %%rax, %[t2]\n\t"
      "mulxq 24(%[a]), %%rax, %[t4]\n\t"
      "adcxq %%rax, %[t3]\n\t"
      "mulxq 32(%[a]), %%rax, %[t5]\n\t"
      "adcxq %%rax, %[t4]\n\t"
      "mulxq 40(%[a]), %%rax, %[A]\n\t"
      "adcxq %%rax, %[t5]\n\t"
      "movl $0, %%eax\n\t"
      "adcxq %%rax, %[A]\n\t"
      NCG_H64_RED_ROW
      NCG_H64_MUL_ROW(8) NCG_H64_RED_ROW
      NCG_H64_MUL_ROW(16) NCG_H64_RED_ROW
      NCG_H64_MUL_ROW(24) NCG_H64_RED_ROW
      NCG_H64_MUL_ROW(32) NCG_H64_RED_ROW
      NCG_H64_MUL_ROW(40) NCG_H64_RED_ROW
      : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [t4] "=&r"(t4), [t5] "=&r"(t5), [A] "=&r"(A), [h] "=&r"(h)
      : [a] "r"(a.v), [b] "r"(b.v), [k] "r"(&k), "m"(a), "m"(b), "m"(k)
      : "rax", "rdx", "cc");
  const uint64_t t[6] = {t0, t1, t2, t3, t4, t5};
  unsigned long long d[6], bw = 0;
  for (int i = 0; i < 6; i++) d[i] = __builtin_subcll(t[i], k.p[i], bw, &bw);
  const uint64_t keep = 0 - (uint64_t)bw;
  Fp r;
  for (int i = 0; i < 6; i++) r.v[i] = (t[i] & keep) | (d[i] & ~keep);
  return r;
}
#undef NCG_H64_MUL_STEP
#undef NCG_H64_MUL_ROW
#undef NCG_H64_RED_STEP
#undef NCG_H64_RED_ROW
inline int& adx_override() {  // test hook (tests/hosttest): 0 forces the portable product, -1 = decide by CPUID
  static int v = -1;
  return v;
}
inline bool have_adx() {
  static const bool ok = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
  return ok && adx_override() != 0;
}
inline Fp mul(const Fp& a, const Fp& b) { return have_adx() ? mul_adx(a, b) : mul_c(a, b); }
#else
#define NCG_H64_ADX 0
inline int& adx_override() {
  static int v = -1;
  return v;
}
inline bool have_adx() { return false; }
inline Fp mul(const Fp& a, const Fp& b) { return mul_c(a, b); }
#endif
// (a separate squaring - 57 products instead of 72 through a 12-word intermediate - measured SLOWER than the fused
// product: 71 ns against 58)
inline Fp sqr(const Fp& a) { return mul(a, a); }
// branch-free: the chain's add / sub are a third of its time when written with compare loops
inline Fp add(const Fp& a, const Fp& b) {
  const Consts& k = K();
  unsigned long long r[6], d[6], c = 0, bw = 0;
  for (int i = 0; i < 6; i++) r[i] = __builtin_addcll(a.v[i], b.v[i], c, &c);   // a + b < 2p < 2^383: no carry out
  for (int i = 0; i < 6; i++) d[i] = __builtin_subcll(r[i], k.p[i], bw, &bw);
  const uint64_t keep = 0 - (uint64_t)bw;  // borrow: r < p, keep r
  Fp o;
  for (int i = 0; i < 6; i++) o.v[i] = (r[i] & keep) | (d[i] & ~keep);
  return o;
}
inline Fp sub(const Fp& a, const Fp& b) {
  const Consts& k = K();
  unsigned long long d[6], bw = 0, c = 0;
  for (int i = 0; i < 6; i++) d[i] = __builtin_subcll(a.v[i], b.v[i], bw, &bw);
  const uint64_t fix = 0 - (uint64_t)bw;   // borrow: add p back
  Fp o;
  for (int i = 0; i < 6; i++) o.v[i] = __builtin_addcll(d[i], k.p[i] & fix, c, &c);
  return o;
}
inline Fp dbl(const Fp& a) { return add(a, a); }
inline Fp neg(const Fp& a) { return is_zero(a) ? a : sub(zero(), a); }
inline Fp one(const Fp*) { return K().one; }
inline Fp inv(const Fp& a) {  // a^(p-2): value of modular.ts:159-182 invert for a != 0
  const Consts& k = K();
  uint64_t e[6];
  memcpy(e, k.p, sizeof e);
  e[0] -= 2;
  Fp r = k.one;
  for (int w = 5; w >= 0; w--)
    for (int bit = 63; bit >= 0; bit--) {
      r = sqr(r);
      if ((e[w] >> bit) & 1) r = mul(r, a);
    }
  return r;
}

// stored device element (14 x 29-bit limbs, R = 2^406, value v below 64 p) -> canonical Montgomery form here (R = 2^384):
// x 2^384 = v 2^-22 mod p.  One Montgomery step with radix 2^22: m = -v p^-1 mod 2^22, (v + m p) >> 22 is below
// 64 p / 2^22 + p, then one conditional subtraction.  (Round 3; the first form canonicalised with a 14-limb product of the
// device template and multiplied by 2^362: 168 ns per element on the build machine against about 20.)
inline Fp from_fe29(const uint32_t* limbs) {
  const Consts& k = K();
  uint64_t v[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 14; i++) {
    const int bit = 29 * i, w = bit >> 6, sh = bit & 63;
    v[w] |= (uint64_t)limbs[i] << sh;
    if (sh > 35 && w + 1 < 7) v[w + 1] |= (uint64_t)limbs[i] >> (64 - sh);
  }
  const uint64_t m = ((0 - v[0]) * k.pinv_pos) & ((1ull << 22) - 1);  // -v / p mod 2^22
  u128 c = 0;
  uint64_t t[7];
  for (int i = 0; i < 6; i++) {
    c += (u128)m * k.p[i] + v[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  c += v[6];
  t[6] = (uint64_t)c;
  Fp r;
  for (int i = 0; i < 6; i++) r.v[i] = (t[i] >> 22) | (t[i + 1] << 42);
  unsigned long long d[6], bw = 0;
  for (int i = 0; i < 6; i++) d[i] = __builtin_subcll(r.v[i], k.p[i], bw, &bw);
  const uint64_t keep = 0 - (uint64_t)bw;
  for (int i = 0; i < 6; i++) r.v[i] = (r.v[i] & keep) | (d[i] & ~keep);
  return r;
}
// -> canonical residue as 12 x 32-bit LE words (the wire format of include/ncg.h)
inline void to_wire(uint32_t* out, const Fp& a) {
  const Fp r = mul(a, K().raw_one);
  for (int i = 0; i < 6; i++) {
    out[2 * i] = (uint32_t)r.v[i];
    out[2 * i + 1] = (uint32_t)(r.v[i] >> 32);
  }
}

// ---- Fp2 = Fp[u] / (u^2 + 1): values of `_Field2` ops, src/abstract/tower.ts:393-475
struct Fp2 {
  Fp c0, c1;
};
inline bool is_zero(const Fp2& a) { return is_zero(a.c0) && is_zero(a.c1); }
inline bool eq(const Fp2& a, const Fp2& b) { return eq(a.c0, b.c0) && eq(a.c1, b.c1); }
inline Fp2 add(const Fp2& a, const Fp2& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
inline Fp2 sub(const Fp2& a, const Fp2& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
inline Fp2 dbl(const Fp2& a) { return {dbl(a.c0), dbl(a.c1)}; }
inline Fp2 mul(const Fp2& a, const Fp2& b) {  // Karatsuba, tower.ts:420-431
  const Fp t1 = mul(a.c0, b.c0), t2 = mul(a.c1, b.c1);
  const Fp m = mul(add(a.c0, a.c1), add(b.c0, b.c1));
  return {sub(t1, t2), sub(m, add(t1, t2))};
}
inline Fp2 sqr(const Fp2& a) {  // tower.ts:432-438
  return {mul(add(a.c0, a.c1), sub(a.c0, a.c1)), mul(dbl(a.c0), a.c1)};
}
inline Fp2 one(const Fp2*) { return {K().one, zero()}; }
inline Fp2 inv(const Fp2& a) {  // tower.ts:458-475
  const Fp f = inv(add(sqr(a.c0), sqr(a.c1)));
  return {mul(f, a.c0), mul(f, neg(a.c1))};
}
inline Fp2 from_fe29x2(const uint32_t* limbs) { return {from_fe29(limbs), from_fe29(limbs + 14)}; }
inline void to_wire(uint32_t* out, const Fp2& a) {
  to_wire(out, a.c0);
  to_wire(out + 12, a.c1);
}
inline Fp from_stored(const uint32_t* p, const Fp*) { return from_fe29(p); }
inline Fp2 from_stored(const uint32_t* p, const Fp2*) { return from_fe29x2(p); }

// ---- Jacobian (X, Y, Z), x = X / Z^2, y = Y / Z^3, infinity Z = 0; a = 0
template <class F>
struct JacH {
  F X, Y, Z;
  bool inf;
};
template <class F>
inline JacH<F> jac_inf() {
  JacH<F> r;
  memset(&r, 0, sizeof r);
  r.inf = true;
  return r;
}
template <class F>
inline JacH<F> jac_dbl(const JacH<F>& p) {  // dbl-2009-l
  if (p.inf) return p;
  const F A = sqr(p.X), B = sqr(p.Y), C = sqr(B);
  const F D = dbl(sub(sub(sqr(add(p.X, B)), A), C));
  const F E = add(dbl(A), A), Fq = sqr(E);
  JacH<F> r;
  r.X = sub(Fq, dbl(D));
  r.Y = sub(mul(E, sub(D, r.X)), dbl(dbl(dbl(C))));
  r.Z = dbl(mul(p.Y, p.Z));
  r.inf = false;  // no point of order 2 on these curves
  return r;
}
template <class F>
inline JacH<F> jac_add(const JacH<F>& p, const JacH<F>& q) {  // add-2007-bl, exceptional cases explicit
  if (q.inf) return p;
  if (p.inf) return q;
  const F Z1Z1 = sqr(p.Z), Z2Z2 = sqr(q.Z);
  const F U1 = mul(p.X, Z2Z2), U2 = mul(q.X, Z1Z1);
  const F S1 = mul(mul(p.Y, q.Z), Z2Z2), S2 = mul(mul(q.Y, p.Z), Z1Z1);
  const F H = sub(U2, U1), rr = dbl(sub(S2, S1));
  if (is_zero(H)) {
    if (is_zero(rr)) return jac_dbl(p);  // P == Q
    return jac_inf<F>();                  // P == -Q
  }
  const F I = sqr(dbl(H)), J = mul(H, I), V = mul(U1, I);
  JacH<F> r;
  r.X = sub(sub(sqr(rr), J), dbl(V));
  r.Y = sub(mul(rr, sub(V, r.X)), dbl(mul(S1, J)));
  r.Z = mul(sub(sub(sqr(add(p.Z, q.Z)), Z1Z1), Z2Z2), H);
  r.inf = false;
  return r;
}
// XYZZ accumulator in device storage (X, Y, ZZ, ZZZ; FW words per coordinate) -> Jacobian (X ZZ, Y ZZZ, ZZ)
template <class F>
inline JacH<F> jac_from_xyzz(const uint32_t* p, int FW) {
  const F ZZ = from_stored(p + 2 * FW, (const F*)nullptr);
  if (is_zero(ZZ)) return jac_inf<F>();
  JacH<F> r;
  r.X = mul(from_stored(p, (const F*)nullptr), ZZ);
  r.Y = mul(from_stored(p + FW, (const F*)nullptr), from_stored(p + 3 * FW, (const F*)nullptr));
  r.Z = ZZ;
  r.inf = false;
  return r;
}

// fin: [ngroups][nwin] grouped sums V_j (msm.hip k_msm_tail): W_w = sum_j 2^(g j) V_j, result = sum_w 2^(c w) W_w.
// out: affine wire (x, y), infinity = (0, 0) + flag.  WW = wire words per coordinate (12 / 24).
template <class F>
inline void jac_to_wire(const JacH<F>& acc, int WW, uint32_t* out, uint8_t* out_inf) {
  memset(out, 0, (size_t)2 * WW * 4);
  *out_inf = acc.inf ? 1 : 0;
  if (acc.inf) return;
  const F zi = inv(acc.Z), zi2 = sqr(zi);
  to_wire(out, mul(acc.X, zi2));
  to_wire(out + WW, mul(mul(acc.Y, zi2), zi));
}
template <class F>
inline void msm_finish_serial(const uint32_t* fin, int c, int nwin, int g, int ngroups, int FW, int WW, uint32_t* out, uint8_t* out_inf) {
  const int XW = 4 * FW;
  JacH<F> acc = jac_inf<F>();
  for (int w = nwin - 1; w >= 0; w--)
    for (int j = ngroups - 1; j >= 0; j--) {
      const int shift = j == ngroups - 1 ? c - g * j : g;
      for (int d = 0; d < shift; d++) acc = jac_dbl(acc);
      acc = jac_add(acc, jac_from_xyzz<F>(fin + ((size_t)j * nwin + w) * XW, FW));
    }
  jac_to_wire(acc, WW, out, out_inf);
}

// ---- the finish over helper threads (round 6) ------------------------------------------------------------------------
// The serial form above spends c doublings AND ngroups complete additions per window on one chain: G2 at c = 13 is 260 doublings
// (16 Fp products each) + 80 additions (49 with the change of representation) - 0.28 ms, a tenth of a 2^18-point MSM and a third
// of a 2^12-point one.  Only the doublings and ONE addition per window have to be on the chain: W_w = sum_j 2^(g j) V_j depends
// on nothing but the window's own ngroups sums, so helper threads build the W_w (top window first, g (ngroups - 1) doublings +
// ngroups - 1 additions each) while the caller runs acc = 2^c acc + W_w behind them.  The doublings cannot be split: the top
// window's sum needs c (nwin - 1) of them one after the other whoever does them.
// The pool: FINISH_HELPERS detached threads asleep on a condition variable.  `wake()` (called while the GPU still runs the
// MSM's tail kernel, msm.hip msm_finish_t) makes them spin for work for at most FINISH_SPIN_US; `run()` publishes the job; windows
// are claimed top-down from one atomic counter by the helpers AND by the caller whenever the window it needs next is not
// there yet, so a job completes whatever the helpers do (asleep, descheduled).  If no helper is spinning when the job arrives the
// caller runs the serial form (building every W_w itself would cost it 75 % more doublings); helpers that have been signalled but
// are still on their way count as present - the caller builds the top windows itself until they arrive.  One job at a time: a second
// caller (another context's thread) finds the pool busy and runs the serial form.
constexpr int FINISH_HELPERS = 3;
constexpr int FINISH_SPIN_US = 600;
struct FinishPool {
  std::mutex busy;                 // one job at a time
  std::mutex m;
  std::condition_variable cv;
  uint64_t wake_gen = 0;           // under m
  std::atomic<int> spinning{0};    // helpers looking for work right now
  std::atomic<int> arriving{0};    // helpers that have been signalled and are not spinning yet (10-50 us from the signal to the first look)
  std::atomic<int> alive{0};       // helper threads that exist
  std::atomic<int> helped{0};      // windows of the current job built by helpers
  int misses = 0, cooldown = 0;    // consecutive jobs no helper took part in; jobs left to run serially after four of those (under
                                   // `busy`): a forked child has no helper threads, a loaded host may have them arrive too late
  std::atomic<int> next{-1};       // next window to claim (counts down); < 0: no job
  void (*fn)(void*, int) = nullptr;  // published before `next`
  void* arg = nullptr;
  std::once_flag started;
  static FinishPool& get() {
    static FinishPool* p = new FinishPool();   // never destroyed: the helpers are detached and outlive static destruction
    return *p;
  }
  void start() {
    std::call_once(started, [this] {
      for (int i = 0; i < FINISH_HELPERS; i++) {
        try {
          std::thread([this] { helper(); }).detach();
          alive.fetch_add(1, std::memory_order_relaxed);
        } catch (...) {   // no helpers: every job runs the serial form
        }
      }
    });
  }
  void helper() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return wake_gen != seen; });
        seen = wake_gen;
      }
      spinning.fetch_add(1, std::memory_order_acq_rel);
      {
        int a = arriving.load(std::memory_order_relaxed);   // one of the signalled helpers has arrived (never below zero: wake-ups can merge)
        while (a > 0 && !arriving.compare_exchange_weak(a, a - 1, std::memory_order_acq_rel)) {
        }
      }
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned it = 0;; it++) {
        if (next.load(std::memory_order_acquire) >= 0) {
          int i;
          while ((i = next.fetch_sub(1, std::memory_order_acq_rel)) >= 0) {
            helped.fetch_add(1, std::memory_order_relaxed);   // before the window is published: the caller reads it once every window is ready
            fn(arg, i);
          }
          break;   // one job per wake-up
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if ((it & 255u) == 255u &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > FINISH_SPIN_US)
          break;
      }
      spinning.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  // make the helpers look for work (a job is about to arrive); cheap when they already are
  void wake() {
    start();
    {
      std::lock_guard<std::mutex> lk(m);
      wake_gen++;
    }
    arriving.store(std::max(0, alive.load(std::memory_order_relaxed) - spinning.load(std::memory_order_relaxed)), std::memory_order_release);
    cv.notify_all();
  }
  // helpers are looking for work, or have just been signalled and will be within tens of microseconds
  bool expected() const { return spinning.load(std::memory_order_acquire) > 0 || arriving.load(std::memory_order_acquire) > 0; }
};
inline int& finish_threads_override() {  // test hook / A-B: 0 = always serial, 1 = helpers whenever they are awake (default), 2 = wake them inside run too
  static int v = 1;
  return v;
}

template <class F>
struct FinishJob {
  const uint32_t* fin;
  int c, nwin, g, ngroups, FW;
  JacH<F>* W;                      // [nwin]
  std::atomic<int>* ready;         // [nwin]
  static void window(void* self, int w) {
    FinishJob& J = *static_cast<FinishJob*>(self);
    const int XW = 4 * J.FW;
    JacH<F> acc = jac_from_xyzz<F>(J.fin + ((size_t)(J.ngroups - 1) * J.nwin + w) * XW, J.FW);
    for (int j = J.ngroups - 2; j >= 0; j--) {
      for (int d = 0; d < J.g; d++) acc = jac_dbl(acc);
      acc = jac_add(acc, jac_from_xyzz<F>(J.fin + ((size_t)j * J.nwin + w) * XW, J.FW));
    }
    J.W[w] = acc;
    J.ready[w].store(1, std::memory_order_release);
  }
};

template <class F>
inline void msm_finish(const uint32_t* fin, int c, int nwin, int g, int ngroups, int FW, int WW, uint32_t* out, uint8_t* out_inf) {
  constexpr int MAXW = 160;
  const int mode = finish_threads_override();
  FinishPool& P = FinishPool::get();
  if (mode == 0 || nwin < 4 || nwin > MAXW || ngroups < 2 || !P.busy.try_lock()) return msm_finish_serial<F>(fin, c, nwin, g, ngroups, FW, WW, out, out_inf);
  std::unique_lock<std::mutex> owner(P.busy, std::adopt_lock);
  if (mode == 2) {
    P.start();
    {
      std::lock_guard<std::mutex> lk(P.m);
      P.wake_gen++;
    }
    P.cv.notify_all();
    for (int spin = 0; spin < 200000 && P.spinning.load(std::memory_order_acquire) == 0; spin++) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  }
  if (P.cooldown > 0 || !P.expected()) {
    if (P.cooldown > 0) P.cooldown--;
    owner.unlock();
    return msm_finish_serial<F>(fin, c, nwin, g, ngroups, FW, WW, out, out_inf);
  }
  JacH<F> W[MAXW];
  std::atomic<int> ready[MAXW];
  for (int w = 0; w < nwin; w++) ready[w].store(0, std::memory_order_relaxed);
  FinishJob<F> J{fin, c, nwin, g, ngroups, FW, W, ready};
  P.fn = &FinishJob<F>::window;
  P.arg = &J;
  P.helped.store(0, std::memory_order_relaxed);
  P.next.store(nwin - 1, std::memory_order_release);
  JacH<F> acc = jac_inf<F>();
  for (int w = nwin - 1; w >= 0; w--) {
    // 2^c acc: the serial form's c - g (ngroups - 1) + g (ngroups - 1) doublings of this window, all of them up front
    for (int d = 0; d < c; d++) acc = jac_dbl(acc);
    while (ready[w].load(std::memory_order_acquire) == 0) {
      // not there yet: take the next unclaimed window ourselves
      const int i = P.next.load(std::memory_order_acquire) >= 0 ? P.next.fetch_sub(1, std::memory_order_acq_rel) : -1;
      if (i >= 0) {
        FinishJob<F>::window(&J, i);
      } else {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
    }
    acc = jac_add(acc, W[w]);
  }
  // every window is ready, so every claim has been served; helpers that still race for `next` find it negative
  P.next.store(-1, std::memory_order_release);
  if (P.helped.load(std::memory_order_relaxed) == 0) {
    if (++P.misses >= 4) {   // signalled four times, never came: run the next 64 finishes serially, then try again
      P.misses = 0;
      P.cooldown = 64;
    }
  } else {
    P.misses = 0;
  }
  owner.unlock();
  jac_to_wire(acc, WW, out, out_inf);
}

}  // namespace h64
}  // namespace ncg
