// ed25519 batch signature verification: one lane per (signature, public key, challenge k).
//
// Reproduces, per item, the boolean of the reference's eddsa.verify (src/abstract/edwards.ts:
// 942-989) for a challenge scalar k = SHA-512(dom || R || A || M) mod L that the host shim has
// already computed (edwards.ts:984, :900-906; the hash lives in @noble/hashes, not on this
// path):
//   A = Point.fromBytes(pk, zip215), R = Point.fromBytes(sig[0:32], zip215)     (:969-970, :405-436)
//   s = LE(sig[32:64]) must be < L (BASE.multiplyUnsafe(s) throws otherwise)       (:962, :971, :573)
//   strict mode only: A.isSmallOrder() rejects                                     (:980)
//   accept iff [8](R + [k]A - [s]B) == O                                           (:985-988)
// The reference computes [s]B with its cached window table (44 adds) and [k]A with a wNAF walk; here all
// streams share ONE doubling chain (Straus), and the default path first halves the scalars (ed_halve.hpp:
// u k = v mod L with |u|, v < 2^127 from a truncated Euclidean algorithm, the verdict is unchanged):
//   default (EdCfgHalf, ed25519_verify_lane_half): accept iff [8]([|u| s mod L]B -+ [v]A - [|u|]R) == O - 128
//     doublings; signed-odd 4-bit windows for -A and -R (two 8-entry per-lane tables of projective Niels points in
//     device memory, 2.3 KB per item), 8-bit windows for the two 128-bit halves of |u| s mod L on the shared tables
//     of B and 2^128 B (2 x 128 precomputed affine Niels multiples): 32 + 32 + 16 + 16 additions, 3 waves/SIMD.
//     2^18 verifications: 2.84 ms against 3.58 ms for the full-size form below (MI355X).
//   full-size scalars, table in device memory (EdCfgGtab, A/B builds only): 4-bit windows for -A, 8-bit for B:
//     264 doublings, 66 + 33 additions.
//   fallback without scratch (EdCfgLds): 2-bit windows for -A (2-entry per-lane table in LDS, projective Niels
//     form: 16 KB per wave, so 8-10 waves fit a CU - a 3-bit window measured 1.7x slower for that reason) and
//     6 bits for B (32 precomputed affine Niels multiples shared by every lane): 258 doublings, 129 + 43 additions.
#include <cstdlib>
#include "knobs.hpp"
#include <mutex>
#include <vector>

#include "ec_te.hpp"
#include "ed_halve.hpp"
#include "host_api.hpp"
#include "scalar.hpp"
#include "sha512.hpp"

namespace ncg {

// Window configuration of the shared doubling chain: WA-bit signed-odd windows for the per-item
// point (table of TA projective Niels entries per lane), WB-bit windows for B (shared affine table
// of 128 entries, the first TB = 2^(WB-1) are used).  WA * MA = WB * MB = BITS >= 254.
//   EdCfgLds  (2, 6): 258 bits - per-lane table in LDS (16 KB per wave)
//   EdCfgGtab (4, 8): 264 bits - per-lane table in device memory (1 KB per item), half the additions
constexpr int ED_FW = FieldIO<FEd>::WORDS;        // stored words per field element (9)
constexpr int ED_NIELS_WORDS = 4 * ED_FW;          // projective Niels entry (Y+X, Y-X, Z, 2dT)
constexpr int ED_AFF_NIELS_WORDS = 3 * ED_FW;      // affine Niels entry (y+x, y-x, 2dxy)
constexpr int ED_PROJ_WORDS = 3 * ED_FW;           // (X, Y, Z) hand-off to the batched inversion
static_assert(ED25519_BTAB_WORDS == 256 * ED_AFF_NIELS_WORDS, "base table size in host_api.hpp");

template <int WA_, int WB_, int BITS_>
struct EdCfg {
  static constexpr int WA = WA_, WB = WB_, BITS = BITS_;
  static constexpr int MA = BITS / WA, MB = BITS / WB;
  static constexpr int TA = 1 << (WA - 1), TB = 1 << (WB - 1);
  static constexpr int LDS_WORDS = TA * ED_NIELS_WORDS * 64;
  static_assert(WA * MA == BITS && WB * MB == BITS && WB % WA == 0 && TB <= 128, "window tiling");
};
using EdCfgLds = EdCfg<2, 6, 258>;
using EdCfgGtab = EdCfg<4, 8, 264>;
// halved scalars (ed_halve.hpp): 128-bit streams for -A and -R, two 128-bit halves of u s mod L on B and 2^128 B
using EdCfgHalf = EdCfg<4, 8, 128>;

NCG_DI bool ed_scalar_lt_L(const uint32_t (&s)[8]) {
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) (void)__builtin_subc(s[i], (uint32_t)Orders::ED_L[i], bw, &bw);
  return bw != 0;
}

template <class PTR>
NCG_DI void ed_store_niels(PTR tab, int stride, int e, const EdNielsProj<FEd>& q) {
  using IO = FieldIO<FEd>;
  constexpr int NW = ED_NIELS_WORDS, FW = ED_FW;
  IO::store_strided(tab + (e * NW + 0) * stride, stride, q.yplusx);
  IO::store_strided(tab + (e * NW + FW) * stride, stride, q.yminusx);
  IO::store_strided(tab + (e * NW + 2 * FW) * stride, stride, q.Z);
  IO::store_strided(tab + (e * NW + 3 * FW) * stride, stride, q.t2d);
}
template <class PTR>
NCG_DI EdNielsProj<FEd> ed_load_niels(PTR tab, int stride, int e) {
  using IO = FieldIO<FEd>;
  constexpr int NW = ED_NIELS_WORDS, FW = ED_FW;
  return {IO::load_strided(tab + (e * NW + 0) * stride, stride), IO::load_strided(tab + (e * NW + FW) * stride, stride),
          IO::load_strided(tab + (e * NW + 2 * FW) * stride, stride),
          IO::load_strided(tab + (e * NW + 3 * FW) * stride, stride)};
}
NCG_DI EdNielsAff<FEd> ed_load_aff_niels(const uint32_t* __restrict__ e) {
  using IO = FieldIO<FEd>;
  return {IO::load(e), IO::load(e + ED_FW), IO::load(e + 2 * ED_FW)};
}
NCG_DI void ed_store_aff_niels(uint32_t* __restrict__ e, const EdNielsAff<FEd>& q) {
  using IO = FieldIO<FEd>;
  IO::store(e, q.yplusx);
  IO::store(e + ED_FW, q.yminusx);
  IO::store(e + 2 * ED_FW, q.t2d);
}
NCG_DI void ed_store_proj(uint32_t* __restrict__ o, const EdExt<FEd>& p) {
  using IO = FieldIO<FEd>;
  IO::store(o, p.X);
  IO::store(o + ED_FW, p.Y);
  IO::store(o + 2 * ED_FW, p.Z);
}

// Per-lane verification; `tab`/`stride` as in mul_var_lane.  btab: 128 affine Niels entries
// [1,3,..,255]B, 24 words each (y+x, y-x, 2dxy).
template <class CFG, class TABPTR>
NCG_DI bool ed25519_verify_lane(const uint32_t* __restrict__ sig, const uint32_t* __restrict__ pk,
                                const uint32_t* __restrict__ kscal, const uint32_t* __restrict__ btab, bool zip215,
                                TABPTR tab, const int stride) {
  using F = FEd;
  static_assert(CFG::BITS >= 254, "full-size scalars");
  uint32_t aw[8], rw[8], s[8], k[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    aw[i] = pk[i];
    rw[i] = sig[i];
    s[i] = sig[8 + i];
    k[i] = kscal[i];
  }
  F ax, ay, rx, ry;
  bool ok = ed_decompress(aw, zip215, ax, ay);
  ok = ed_decompress(rw, zip215, rx, ry) && ok;
  ok = ok && ed_scalar_lt_L(s);
  const F d2 = EdConsts::d2();

  // -A in extended coordinates, its double, and the table [1,3,5,7](-A) in Niels form
  EdExt<F> nA{f_neg(ax), ay, F::one(), f_neg(ax * ay)};
  if (!zip215) {  // strict: reject small-order A  (isSmallOrder: [8]A == O)
    EdExt<F> t = ed_dbl(ed_dbl(ed_dbl(nA)));
    if (ed_is_identity(t)) ok = false;
  }
  {
    EdNielsProj<F> n2 = ed_to_niels(ed_dbl(nA), d2);
    EdExt<F> cur = nA;
    ed_store_niels(tab, stride, 0, ed_to_niels(cur, d2));
#pragma unroll
    for (int j = 1; j < CFG::TA; j++) {
      cur = ed_add_niels(cur, n2, false);
      ed_store_niels(tab, stride, j, ed_to_niels(cur, d2));
    }
  }
  SignedOddWindows<9, CFG::WA, CFG::MA> wk;
  SignedOddWindows<9, CFG::WB, CFG::MB> ws;
  wk.template init<8>(k);
  ws.template init<8>(s);

  EdExt<F> acc = EdExt<F>::identity();
  for (int i = CFG::MA - 1; i >= 0; i--) {
    if (i != CFG::MA - 1) {
#pragma unroll(NCG_MUL_INLINE ? 1 : CFG::WA)
      for (int d = 0; d < CFG::WA - 1; d++) acc = ed_dbl_no_t(acc);
      acc = ed_dbl(acc);
    }
    int dA = wk.pop();
    acc = ed_add_niels(acc, ed_load_niels(tab, stride, ((dA < 0 ? -dA : dA) - 1) >> 1), dA < 0);
    if (i % (CFG::WB / CFG::WA) == 0) {
      int dB = ws.pop();
      acc = ed_madd_niels(acc, ed_load_aff_niels(btab + (((dB < 0 ? -dB : dB) - 1) >> 1) * ED_AFF_NIELS_WORDS), dB < 0);
    }
  }
  if (wk.was_even) acc = ed_add_niels(acc, ed_load_niels(tab, stride, 0), true);
  if (ws.was_even) acc = ed_madd_niels(acc, ed_load_aff_niels(btab), true);
  // acc = [s]B - [k]A ; subtract R, clear the cofactor, compare with the identity
  acc = ed_madd_niels(acc, ed_affine_to_niels(rx, ry, d2), true);
  acc = ed_dbl(ed_dbl(ed_dbl(acc)));
  return ok && ed_is_identity(acc);
}

// The same verdict from half-size scalars (ed_halve.hpp): (u, v) with u k == v (mod L), then
//     accept iff [8]([|u| s mod L] B -+ [v] A - [|u|] R) == O        (upper sign for u > 0)
// on ONE 128-step doubling chain: 4-bit signed-odd windows for -A and -R (two per-lane tables of TA projective
// Niels entries), 8-bit windows for the two halves of |u| s mod L on the shared tables of B and 2^128 B
// (btab: 2 x 128 affine Niels entries).  128 doublings and 32 + 32 + 16 + 16 additions instead of 264 and 66 + 33.
template <class CFG, class TABPTR>
NCG_DI bool ed25519_verify_lane_half(const uint32_t* __restrict__ sig, const uint32_t* __restrict__ pk,
                                     const uint32_t* __restrict__ kscal, const uint32_t* __restrict__ btab, bool zip215,
                                     TABPTR tab, const int stride) {
  using F = FEd;
  static_assert(CFG::BITS == 128 && CFG::WB == 2 * CFG::WA, "two windows of the point streams per base window");
  uint32_t aw[8], rw[8], s[8], k[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    aw[i] = pk[i];
    rw[i] = sig[i];
    s[i] = sig[8 + i];
    k[i] = kscal[i];
  }
  F ax, ay, rx, ry;
  bool ok = ed_decompress(aw, zip215, ax, ay);
  ok = ed_decompress(rw, zip215, rx, ry) && ok;
  ok = ok && ed_scalar_lt_L(s);
  const F d2 = EdConsts::d2();
  // tables [1,3,..,2 TA - 1] (-A) at entries 0.., [1,3,..](-R) at entries TA..
  auto build = [&](const F& x, const F& y, int base, bool check_small) {
    EdExt<F> nP{f_neg(x), y, F::one(), f_neg(x * y)};
    EdExt<F> dbl = ed_dbl(nP);
    if (check_small) {  // strict: reject small-order A  (isSmallOrder: [8]A == O)
      if (ed_is_identity(ed_dbl(ed_dbl(dbl)))) ok = false;
    }
    EdNielsProj<F> n2 = ed_to_niels(dbl, d2);
    EdExt<F> cur = nP;
    ed_store_niels(tab, stride, base, ed_to_niels(cur, d2));
#pragma unroll 1
    for (int j = 1; j < CFG::TA; j++) {
      cur = ed_add_niels(cur, n2, false);
      ed_store_niels(tab, stride, base + j, ed_to_niels(cur, d2));
    }
  };
  build(ax, ay, 0, !zip215);
  build(rx, ry, CFG::TA, false);
  const EdHalf hv = ed_halve_scalar(k);
  uint32_t w[8];
  ed_mul_mod_l(w, hv.u, s);
  const uint32_t wlo[4] = {w[0], w[1], w[2], w[3]}, whi[4] = {w[4], w[5], w[6], w[7]};
  SignedOddWindows<4, CFG::WA, CFG::MA> wa, wr;
  SignedOddWindows<4, CFG::WB, CFG::MB> wb0, wb1;
  wa.template init<4>(hv.v);
  wr.template init<4>(hv.u);
  wb0.template init<4>(wlo);
  wb1.template init<4>(whi);
  const uint32_t* btab1 = btab + 128 * ED_AFF_NIELS_WORDS;

  EdExt<F> acc = EdExt<F>::identity();
#pragma unroll 1
  for (int i = CFG::MA - 1; i >= 0; i--) {
    if (i != CFG::MA - 1) {
#pragma unroll(NCG_MUL_INLINE ? 1 : CFG::WA)
      for (int d = 0; d < CFG::WA - 1; d++) acc = ed_dbl_no_t(acc);
      acc = ed_dbl(acc);
    }
    // the additions of the two point streams (-A, -R) are one loop body run twice, and so are the two of the base
    // streams: the window body is 40 KB of code instead of 62 KB (the instruction cache holds 64 KB)
#pragma unroll 1
    for (int e = 0; e < 2; e++) {
      const int dP = e == 0 ? wa.pop() : wr.pop();
      const bool ngP = e == 0 ? ((dP < 0) != hv.uneg) : (dP < 0);
      const int ent = (e == 0 ? 0 : CFG::TA) + (((dP < 0 ? -dP : dP) - 1) >> 1);
      acc = ed_add_niels(acc, ed_load_niels(tab, stride, ent), ngP);
    }
    if ((i & 1) == 0) {
#pragma unroll 1
      for (int e = 0; e < 2; e++) {
        const int dB = e == 0 ? wb0.pop() : wb1.pop();
        const uint32_t* bt = e == 0 ? btab : btab1;
        acc = ed_madd_niels(acc, ed_load_aff_niels(bt + (((dB < 0 ? -dB : dB) - 1) >> 1) * ED_AFF_NIELS_WORDS), dB < 0);
      }
    }
  }
  // even scalars were bumped by one: take the extra points back out
  if (wa.was_even) acc = ed_add_niels(acc, ed_load_niels(tab, stride, 0), !hv.uneg);
  if (wr.was_even) acc = ed_add_niels(acc, ed_load_niels(tab, stride, CFG::TA), true);
  if (wb0.was_even) acc = ed_madd_niels(acc, ed_load_aff_niels(btab), true);
  if (wb1.was_even) acc = ed_madd_niels(acc, ed_load_aff_niels(btab1), true);
  acc = ed_dbl(ed_dbl(ed_dbl(acc)));
  return ok && ed_is_identity(acc);
}

// GTAB: per-lane table in device memory at gtab + idx * TA * 32 (item-major, stride 1), no LDS
template <class CFG, bool GTAB, int MINW>
__global__ void __launch_bounds__(64, MINW)
k_ed25519_verify(const uint32_t* __restrict__ sigs, const uint32_t* __restrict__ pks,
                 const uint32_t* __restrict__ ks, const uint32_t* __restrict__ btab, int zip215,
                 uint8_t* __restrict__ out_ok, uint32_t* __restrict__ gtab, int n) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x;
  const int idx = blockIdx.x * 64 + lane;
  const int src = idx < n ? idx : n - 1;
  bool ok;
  if constexpr (GTAB)
    ok = ed25519_verify_lane<CFG>(sigs + (size_t)src * 16, pks + (size_t)src * 8, ks + (size_t)src * 8, btab, zip215 != 0,
                                  gtab + (size_t)idx * (CFG::TA * ED_NIELS_WORDS), 1);
  else
    ok = ed25519_verify_lane<CFG>(sigs + (size_t)src * 16, pks + (size_t)src * 8, ks + (size_t)src * 8, btab, zip215 != 0,
                                  lds + lane, 64);
  if (idx < n) out_ok[idx] = ok ? 1 : 0;
}

// halved scalars: two per-lane tables (-A, -R) in device memory at gtab + idx * 2 TA * 36
template <class CFG, int MINW>
__global__ void __launch_bounds__(64, MINW)
k_ed25519_verify_half(const uint32_t* __restrict__ sigs, const uint32_t* __restrict__ pks, const uint32_t* __restrict__ ks,
                      const uint32_t* __restrict__ btab, int zip215, uint8_t* __restrict__ out_ok,
                      uint32_t* __restrict__ gtab, int n) {
  const int idx = blockIdx.x * 64 + threadIdx.x;
  const int src = idx < n ? idx : n - 1;
  const bool ok = ed25519_verify_lane_half<CFG>(sigs + (size_t)src * 16, pks + (size_t)src * 8, ks + (size_t)src * 8, btab,
                                                zip215 != 0, gtab + (size_t)idx * (2 * CFG::TA * ED_NIELS_WORDS), 1);
  if (idx < n) out_ok[idx] = ok ? 1 : 0;
}

// ---- batch variable-base multiplication on ed25519: out[i] = k[i] * P[i] ---------------------
// Point.multiplyUnsafe / the value of Point.multiply (src/abstract/edwards.ts:555-577 ->
// wnaf.mulUnsafe -> mulAddUnsafe, src/abstract/curve.ts:752-764).  Exact integer multiples for
// any curve point, torsion components included (test/ed25519.test.ts:355-390): the signed-odd
// recoding is an identity over the integers.  Wire points are affine (x, y); the identity is (0, 1).
// PROJ_OUT: write (X, Y, Z) (8 words each) to out_wire and leave the inversion to
// k_ed_batch_affine (one inversion per 8 points instead of one ~265-multiplication chain per lane).
template <class CFG, bool PROJ_OUT = false, class TABPTR>
NCG_DI void ed25519_mul_var_lane(const uint32_t* __restrict__ pt_wire, const uint32_t* __restrict__ k_wire,
                                 uint32_t* __restrict__ out_wire, uint8_t* __restrict__ out_inf, bool active,
                                 TABPTR tab, const int stride) {
  using F = FEd;
  F x = FieldWire<F>::load(pt_wire), y = FieldWire<F>::load(pt_wire + 8);
  uint32_t k[8];
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = k_wire[i];
  const bool kzero = mp_is_zero<8>(k);
  const F d2 = EdConsts::d2();
  EdExt<F> P{x, y, F::one(), x * y};
  {
    EdNielsProj<F> n2 = ed_to_niels(ed_dbl(P), d2);
    EdExt<F> cur = P;
    ed_store_niels(tab, stride, 0, ed_to_niels(cur, d2));
#pragma unroll
    for (int j = 1; j < CFG::TA; j++) {
      cur = ed_add_niels(cur, n2, false);
      ed_store_niels(tab, stride, j, ed_to_niels(cur, d2));
    }
  }
  SignedOddWindows<9, CFG::WA, CFG::MA> wk;
  wk.template init<8>(k);
  EdExt<F> acc = EdExt<F>::identity();
  for (int i = CFG::MA - 1; i >= 0; i--) {
    if (i != CFG::MA - 1) {
#pragma unroll
      for (int d = 0; d < CFG::WA - 1; d++) acc = ed_dbl_no_t(acc);
      acc = ed_dbl(acc);
    }
    int dA = wk.pop();
    acc = ed_add_niels(acc, ed_load_niels(tab, stride, ((dA < 0 ? -dA : dA) - 1) >> 1), dA < 0);
  }
  if (wk.was_even) acc = ed_add_niels(acc, ed_load_niels(tab, stride, 0), true);
  if (kzero) acc = EdExt<F>::identity();
  if constexpr (PROJ_OUT) {
    if (active) ed_store_proj(out_wire, acc);
    return;
  }
  F zi = f_inv(acc.Z);
  F ox = acc.X * zi, oy = acc.Y * zi;
  if (active) {
    FieldWire<F>::store(out_wire, ox);
    FieldWire<F>::store(out_wire + 8, oy);
    *out_inf = (f_eqz(ox) && f_eq(oy, F::one())) ? 1 : 0;
  }
}

template <class CFG, bool PROJ_OUT, bool GTAB, int MINW>
__global__ void __launch_bounds__(64, MINW)
k_ed25519_mul_var(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ scalars, uint32_t* __restrict__ out,
                  uint8_t* __restrict__ out_inf, uint32_t* __restrict__ gtab, int n) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x;
  const int idx = blockIdx.x * 64 + lane;
  const bool active = idx < n;
  const int src = active ? idx : n - 1;
  if constexpr (GTAB)
    ed25519_mul_var_lane<CFG, PROJ_OUT>(pts + (size_t)src * 16, scalars + (size_t)src * 8,
                                        out + (size_t)src * (PROJ_OUT ? ED_PROJ_WORDS : 16), out_inf + src, active,
                                        gtab + (size_t)idx * (CFG::TA * ED_NIELS_WORDS), 1);
  else
    ed25519_mul_var_lane<CFG, PROJ_OUT>(pts + (size_t)src * 16, scalars + (size_t)src * 8,
                                        out + (size_t)src * (PROJ_OUT ? ED_PROJ_WORDS : 16), out_inf + src, active, lds + lane, 64);
}

// (X, Y, Z) -> affine wire (x, y) = (X/Z, Y/Z) with Montgomery's trick over K consecutive points
// (toAffine edwards.ts:595-609 / FpInvertBatch modular.ts:728-760); flag = 1 for the identity (0, 1)
template <int K>
__global__ void __launch_bounds__(256) k_ed_batch_affine(const uint32_t* __restrict__ proj, uint32_t* __restrict__ out_wire,
                                                         uint8_t* __restrict__ out_inf, int n) {
  using F = FEd;
  using IO = FieldIO<F>;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i0 = t * K;
  if (i0 >= n) return;
  F pre[K];
  F acc = F::one();
#pragma unroll
  for (int j = 0; j < K; j++) {
    pre[j] = acc;
    if (i0 + j < n) acc = acc * IO::load(proj + (size_t)(i0 + j) * ED_PROJ_WORDS + 2 * ED_FW);  // Z != 0 on Edwards curves
  }
  F inv = f_inv(acc);
#pragma unroll
  for (int j = K - 1; j >= 0; j--) {
    if (i0 + j < n) {
      const uint32_t* p = proj + (size_t)(i0 + j) * ED_PROJ_WORDS;
      F z = IO::load(p + 2 * ED_FW);
      F zi = inv * pre[j];
      inv = inv * z;
      F x = IO::load(p) * zi, y = IO::load(p + ED_FW) * zi;
      FieldWire<F>::store(out_wire + (size_t)(i0 + j) * 16, x);
      FieldWire<F>::store(out_wire + (size_t)(i0 + j) * 16 + 8, y);
      out_inf[i0 + j] = (f_eqz(x) && f_eq(y, F::one())) ? 1 : 0;
    }
  }
}

// (X, Y, Z) triples (ED_PROJ_WORDS words each, as written by the kernels above) -> affine wire points + identity flags
hipError_t ed25519_proj_to_affine(const uint32_t* proj, uint32_t* out, uint8_t* out_inf, int n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_ed_batch_affine<8>, dim3(((n + 7) / 8 + 255) / 256), dim3(256), 0, st, proj, out, out_inf, n);
  return hipGetLastError();
}

// proj_tmp: ed25519_tmp_words(n) words of device scratch ((X, Y, Z) per item + the per-item window
// tables), or nullptr (LDS table, per-lane inversion)
size_t ed25519_tmp_words(int n) { return ((size_t)n + 63) / 64 * 64 * (ED_PROJ_WORDS + EdCfgGtab::TA * ED_NIELS_WORDS); }
hipError_t ed25519_mul_var_batch(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                                 uint32_t* proj_tmp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (proj_tmp) {
    uint32_t* gtab = proj_tmp + ((size_t)n + 63) / 64 * 64 * ED_PROJ_WORDS;
    hipLaunchKernelGGL((k_ed25519_mul_var<EdCfgGtab, true, true, 4>), dim3((n + 63) / 64), dim3(64), 0, st, pts, scalars,
                       proj_tmp, out_inf, gtab, n);
    hipLaunchKernelGGL(k_ed_batch_affine<8>, dim3(((n + 7) / 8 + 255) / 256), dim3(256), 0, st, proj_tmp, out, out_inf, n);
    return hipGetLastError();
  }
  size_t lds = (size_t)EdCfgLds::LDS_WORDS * 4;
  auto kern = k_ed25519_mul_var<EdCfgLds, false, false, 1>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((n + 63) / 64), dim3(64), lds, st, pts, scalars, out, out_inf, (uint32_t*)nullptr, n);
  return hipGetLastError();
}

// ---- batch fixed-base multiplication out[i] = k[i] * BASE ---------------------------------------
// BASE.multiply(k) through the cached window table of the reference (wnafCachedCT,
// src/abstract/curve.ts:588-606; e.g. getPublicKey): signed-odd 8-bit windows, table[w][j] =
// (2j+1) 2^(8w) B as affine Niels points (33 x 128 x 96 B = 405 KB, L2-resident), so one multiply
// is 33 mixed additions (7M each) and no doublings.
constexpr int ED_FB_W = 8, ED_FB_M = 33, ED_FB_T = 1 << (ED_FB_W - 1);
__global__ void __launch_bounds__(256) k_ed_mul_base(const uint32_t* __restrict__ table, const uint32_t* __restrict__ scalars,
                                                     uint32_t* __restrict__ proj_out, int n) {
  using F = FEd;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k[8];
#pragma unroll
  for (int j = 0; j < 8; j++) k[j] = scalars[(size_t)i * 8 + j];
  const bool zero = mp_is_zero<8>(k);
  SignedOddWindows<9, ED_FB_W, ED_FB_M> win;
  win.template init<8>(k);
  EdExt<F> acc = EdExt<F>::identity();
  for (int w = ED_FB_M - 1; w >= 0; w--) {
    const int d = win.pop();
    const uint32_t* e = table + ((size_t)w * ED_FB_T + (((d < 0 ? -d : d) - 1) >> 1)) * ED_AFF_NIELS_WORDS;
    acc = ed_madd_niels(acc, ed_load_aff_niels(e), d < 0);
  }
  if (win.was_even) acc = ed_madd_niels(acc, ed_load_aff_niels(table), true);  // the scalar was bumped by one: take BASE back out
  if (zero) acc = EdExt<F>::identity();
  ed_store_proj(proj_out + (size_t)i * ED_PROJ_WORDS, acc);
}

hipError_t ed25519_mul_base_batch(const uint32_t* table, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                                  uint32_t* proj_tmp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_ed_mul_base, dim3((n + 255) / 256), dim3(256), 0, st, table, scalars, proj_tmp, n);
  hipLaunchKernelGGL(k_ed_batch_affine<8>, dim3(((n + 7) / 8 + 255) / 256), dim3(256), 0, st, proj_tmp, out, out_inf, n);
  return hipGetLastError();
}

void ed25519_mul_var_host(const uint32_t* pt, const uint32_t* k, uint32_t* out, uint8_t* out_inf) {
  std::vector<uint32_t> tab(EdCfgGtab::TA * ED_NIELS_WORDS);
  ed25519_mul_var_lane<EdCfgGtab, false>(pt, k, out, out_inf, true, tab.data(), 1);
}

static EdExt<FEd> ed_base_point() {  // src/ed25519.ts:57-65 Gx, Gy
  static const uint32_t GX[8] = {0x8f25d51au, 0xc9562d60u, 0x9525a7b2u, 0x692cc760u, 0xfdd6dc5cu, 0xc0a4e231u,
                                 0xcd6e53feu, 0x216936d3u};
  static const uint32_t GY[8] = {0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u,
                                 0x66666666u, 0x66666666u};
  FEd gx = FieldWire<FEd>::load(GX), gy = FieldWire<FEd>::load(GY);
  return {gx, gy, FEd::one(), gx * gy};
}

// fixed-base table for k_ed_mul_base, computed on the host with the same templates: 33 x 128
// entries of 24 words, entry [w][j] = (2j+1) 2^(8w) B; one batched inversion for all 4224 points
size_t ed25519_fixed_table_words() { return (size_t)ED_FB_M * ED_FB_T * ED_AFF_NIELS_WORDS; }
void ed25519_build_fixed_table(uint32_t* out) {
  using F = FEd;
  const F d2 = EdConsts::d2();
  const int NT = ED_FB_M * ED_FB_T;
  std::vector<EdExt<F>> pts(NT);
  EdExt<F> bw = ed_base_point();
  for (int w = 0; w < ED_FB_M; w++) {
    EdNielsProj<F> n2 = ed_to_niels(ed_dbl(bw), d2);
    EdExt<F> cur = bw;
    for (int j = 0; j < ED_FB_T; j++) {
      if (j > 0) cur = ed_add_niels(cur, n2, false);
      pts[(size_t)w * ED_FB_T + j] = cur;
    }
    for (int d = 0; d < ED_FB_W; d++) bw = ed_dbl(bw);
  }
  std::vector<F> pre(NT);
  F acc = F::one();
  for (int i = 0; i < NT; i++) {
    pre[i] = acc;
    acc = acc * pts[i].Z;
  }
  F inv = f_inv(acc);
  for (int i = NT - 1; i >= 0; i--) {
    F zi = inv * pre[i];
    inv = inv * pts[i].Z;
    F x = pts[i].X * zi, y = pts[i].Y * zi;
    ed_store_aff_niels(out + (size_t)i * ED_AFF_NIELS_WORDS, ed_affine_to_niels(x, y, d2));
  }
}

// ---- base-point tables [1,3,..,255] B and [1,3,..,255] (2^128 B) in affine Niels form (host-computed with the
// same templates); the second half serves the high half of the halved-scalar verification's fixed-base scalar
void ed25519_build_base_table(uint32_t* out /* ED25519_BTAB_WORDS */) {
  using F = FEd;
  const F d2 = EdConsts::d2();
  EdExt<F> B = ed_base_point();
  for (int half = 0; half < 2; half++) {
    EdNielsProj<F> n2 = ed_to_niels(ed_dbl(B), d2);
    EdExt<F> cur = B;
    for (int j = 0; j < 128; j++) {
      if (j > 0) cur = ed_add_niels(cur, n2, false);
      F zi = f_inv(cur.Z);
      F x = cur.X * zi, y = cur.Y * zi;
      ed_store_aff_niels(out + (half * 128 + j) * ED_AFF_NIELS_WORDS, ed_affine_to_niels(x, y, d2));
    }
    for (int d = 0; d < 128; d++) B = ed_dbl(B);
  }
}

// gtab: ed25519_verify_tmp_words(n) words of device scratch for the per-item tables, or nullptr (LDS variant, full-size
// scalars)
size_t ed25519_verify_tmp_words(int n) { return ((size_t)n + 63) / 64 * 64 * 2 * EdCfgHalf::TA * ED_NIELS_WORDS; }
hipError_t ed25519_verify_batch(const uint32_t* sigs, const uint32_t* pks, const uint32_t* ks, const uint32_t* btab,
                                int zip215, uint8_t* out_ok, int n, uint32_t* gtab, hipStream_t st) {
  if (n <= 0) return hipSuccess;
#ifdef NCG_AB_BUILD
  static const int variant = knob("NCG_ED_VARIANT", 5);
#else
  constexpr int variant = 5;  // halved scalars, 3 waves/SIMD; 2-4: full-size scalars at 2-4 waves/SIMD (A/B builds)
#endif
  if (gtab && variant > 0) {
#ifdef NCG_AB_BUILD
    if (variant == 2)
      hipLaunchKernelGGL((k_ed25519_verify<EdCfgGtab, true, 2>), dim3((n + 63) / 64), dim3(64), 0, st, sigs, pks, ks, btab,
                         zip215, out_ok, gtab, n);
    else if (variant == 3)
      hipLaunchKernelGGL((k_ed25519_verify<EdCfgGtab, true, 3>), dim3((n + 63) / 64), dim3(64), 0, st, sigs, pks, ks, btab,
                         zip215, out_ok, gtab, n);
    else if (variant == 4)
      hipLaunchKernelGGL((k_ed25519_verify<EdCfgGtab, true, 4>), dim3((n + 63) / 64), dim3(64), 0, st, sigs, pks, ks, btab,
                         zip215, out_ok, gtab, n);
    else if (variant == 6)
      hipLaunchKernelGGL((k_ed25519_verify_half<EdCfgHalf, 2>), dim3((n + 63) / 64), dim3(64), 0, st, sigs, pks, ks, btab, zip215,
                         out_ok, gtab, n);
    else if (variant == 7)
      hipLaunchKernelGGL((k_ed25519_verify_half<EdCfgHalf, 4>), dim3((n + 63) / 64), dim3(64), 0, st, sigs, pks, ks, btab, zip215,
                         out_ok, gtab, n);
    else
#endif
      hipLaunchKernelGGL((k_ed25519_verify_half<EdCfgHalf, 3>), dim3((n + 63) / 64), dim3(64), 0, st, sigs, pks, ks, btab, zip215,
                         out_ok, gtab, n);
    return hipGetLastError();
  }
  size_t lds = (size_t)EdCfgLds::LDS_WORDS * 4;
  auto kern = k_ed25519_verify<EdCfgLds, false, 1>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((n + 63) / 64), dim3(64), lds, st, sigs, pks, ks, btab, zip215, out_ok, (uint32_t*)nullptr, n);
  return hipGetLastError();
}

// ---- challenge scalars k = SHA-512(R || A || M) mod L for a batch (sha512.hpp), one lane per signature.
// msgs: all messages back to back; msg_off: n + 1 byte offsets into it (message i = [off[i], off[i+1])).
__global__ void __launch_bounds__(256) k_ed25519_challenge(const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ pks,
                                                           const uint8_t* __restrict__ msgs,
                                                           const uint64_t* __restrict__ msg_off, uint32_t* __restrict__ ks,
                                                           int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t lo = msg_off[i], hi = msg_off[i + 1];
  uint64_t h[8];
  sha512_ram(h, sigs + (size_t)i * 64, pks + (size_t)i * 32, msgs + lo, hi - lo);
  uint32_t k[8];
  sha512_digest_mod_l(k, h);
#pragma unroll
  for (int j = 0; j < 8; j++) ks[(size_t)i * 8 + j] = k[j];
}
hipError_t ed25519_challenge_batch(const uint8_t* sigs, const uint8_t* pks, const uint8_t* msgs, const uint64_t* msg_off,
                                   uint32_t* ks, int n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_ed25519_challenge, dim3((n + 255) / 256), dim3(256), 0, st, sigs, pks, msgs, msg_off, ks, n);
  return hipGetLastError();
}
void ed25519_challenge_host(const uint8_t* sig, const uint8_t* pk, const uint8_t* msg, uint64_t len, uint32_t* k_out) {
  uint64_t h[8];
  sha512_ram(h, sig, pk, msg, len);
  uint32_t k[8];
  sha512_digest_mod_l(k, h);
  for (int j = 0; j < 8; j++) k_out[j] = k[j];
}

// host-only: run the lane function on the CPU (unit tests through hosttest.hip)
bool ed25519_verify_host(const uint32_t* sig, const uint32_t* pk, const uint32_t* k, const uint32_t* btab, bool zip215) {
  std::vector<uint32_t> tab(2 * EdCfgHalf::TA * ED_NIELS_WORDS);
  return ed25519_verify_lane_half<EdCfgHalf>(sig, pk, k, btab, zip215, tab.data(), 1);
}

}  // namespace ncg
