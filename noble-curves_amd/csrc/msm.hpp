// Pippenger multi-scalar multiplication sum_i s_i * P_i on one GPU.
//
// Reproduces the result of the reference's `pippenger(c, points, scalars)`
// (src/abstract/curve.ts:863-905).  The reference walks unsigned c-bit windows serially,
// adds every point into `buckets[digit]` with complete projective additions (:890-894),
// folds the buckets with a running sum (:897-900) and Horner-combines windows (:901-902).
// Only the final group element is contractual (canonical affine), so the GPU
// pipeline is organised for the machine instead:
//
//   1. k_points_to_mont   affine wire points -> Montgomery limbs, once            (HBM stream)
//   2. k_msm_digits       signed c-bit digits for every (scalar, window): add the constant
//                         H' = sum 2^(c-1) 2^(cw) once, then every window is an independent
//                         bit-field minus 2^(c-1)  (digits in [-2^(c-1), 2^(c-1)-1]; signed
//                         digits halve the bucket count - an integer identity, always exact)
//   3. k_msm_hist / k_msm_scan / k_msm_scatter   counting sort of point indices by
//                         (window, |digit|): per-block LDS histograms (2^(c-1) counters, 128 KB
//                         at c = 16), one scan block per window, LDS-atomic scatter
//   4. k_msm_accum        bucket sums: XYZZ accumulator += affine point (mixed add, 10 field
//                         muls instead of the reference's 14), points gathered by sorted index
//   5. k_msm_reduce_level bucket fold sum_b b*B_b as log2(#buckets) pairwise levels
//                         (S' = S[2q]+S[2q+1], R = S[2q+1]; pending R-arrays are pair-summed),
//                         depth log instead of the reference's serial running sum
//   6. host               the c*nwin surviving points are Horner-combined (256 doublings: a
//                         serial chain that a latency-optimised core does ~30x faster than one
//                         GPU lane) and normalised to affine (curve.ts:311-326 / toAffine)
//
// Zero digits are skipped (adding into bucket 0 never reaches the reference's result either,
// curve.ts:896 "Skip first bucket"), infinity inputs are skipped, P = Q / P = -Q inside a
// bucket are handled by the group-law routines (ec_sw.hpp).
#pragma once
#include "curves.hpp"
#include "host_api.hpp"

namespace ncg {

struct MsmTrace;
struct MsmPlan {
  int n = 0;       // points
  int c = 0;       // window bits
  int nwin = 0;    // windows
  int nb = 0;      // buckets per window = 2^(c-1)
  int Q = 0;       // sort chunks (blocks per window)
  int chunk = 0;   // points per chunk
  int ls = 0;      // lanes per item = 1 << ls (lane-paired Fp2 kernels)
  int accum_waves = 2;  // waves/SIMD of the accumulate kernel (sets the resident-lane capacity)
  uint32_t hconst[10];  // H' = sum_w 2^(c-1) * 2^(c w), 10 LE limbs
  uint32_t order[8];    // group order: scalars must be below it (the window plan covers (order-1) + H')
  // endomorphism mode (endo.hpp; point sets verified to lie in the prime-order subgroup): every scalar is
  // split into `endo` balanced sub-scalars (2 x 128 bits on G1, 4 x 64 bits on G2) that multiply the point
  // and its endomorphism images; n = endo * n_src entries, the windows cover the sub-scalar width only
  int endo = 0;
  int n_src = 0;
  int xcd_map = 0;  // sort kernels: window-major block ids so that a window's blocks share an XCD (one L2)
  int scatter_passes = 1;  // k_msm_scatter launches, one bucket range each (with xcd_map: the range a pass writes stays in L2)
  int pts_stored = 0;  // the caller's points are already in the accumulate kernel's storage format (resident sets)
  // shared-bucket mode (precomputed sets, msm_precomp.hip): the point array holds one window-shifted copy of the
  // set per window, level w = 2^(c w) P at [w * n, (w + 1) * n), so every window adds into ONE bucket set: sorted
  // entry = w * n + i, and everything after the sort runs as a single window of nwin * n entries
  int shared = 0;
  // window subset (window-sharded multi-GPU mode, comm.hip): this plan runs windows [w0, w0 + nwin) of a full plan of
  // nwin_total windows; the digit kernels cut bit position c (w0 + w), everything after them sees nwin local windows
  int w0 = 0;
  int nwin_total = 0;  // 0 = the plan is whole (nwin_total == nwin)
  // SHORT TOP WINDOW (round 5): when the plan's last window holds few bits (255 = 19 x 13 + 8; secp256k1 at c = 16: 1-2 bits) its
  // digits 1 .. 2^top_tb would put every entry of the window into a handful of buckets - runs of thousands of pieces for the fix-up.
  // The digit kernel then spreads them: bucket = (index & top_submask) << top_tb | (digit - 1), i.e. 2^(c-1-top_tb) sub-buckets
  // per digit value, and the tail drops the fold's pending sums of the levels >= top_tb for that window: the weight of bucket b
  // becomes (b mod 2^top_tb) + 1 = the digit.  0 = off (full top window, endomorphism and shared-bucket plans).
  int top_tb = 0;
  uint32_t top_submask = 0;
  // per-context tuning overrides (ncg_msm_set_tuning; 0 / -1 = the measured defaults): entries per accumulate lane, and
  // how many following pieces the owner of a cut bucket adds itself before the run goes to the work list
  int seg_override = 0;
  int run_serial_override = -1;
  MsmTrace* trace = nullptr;  // host side: what the last launch actually used (ncg_msm_last_plan)
  // The points of ONE MSM in several parts (the host-pointer entry point: part p is accumulated while part p + 1 is still
  // crossing PCIe): every part runs digits / sort / accumulate / fix-up on ITS points - same c, nwin and bucket arrays -
  // and a bucket's accumulator starts from what the earlier parts left in it instead of the identity; the fold and the
  // tail run once, after the last part.  part_flags: bit 0 = first part (clears the buckets and the scalar verdict),
  // bit 1 = last part (fold + tail).  n_layout: the workspace layout (and the lane segment) are those of an n_layout-point
  // plan, so that all parts address the same bucket array; index_base: global index of the part's first point.
  int part_flags = 3;
  int n_layout = 0;
  int Q_layout = 0;   // sort chunks of the layout plan (the per-chunk count arrays are sized by it)
  uint32_t index_base = 0;
  // host side (msm_run): a word of pinned host memory that k_msm_tail sets to tail_gen when it STARTS - the caller then wakes the
  // helper threads of the host finish (bls_host64.hpp FinishPool) while the tail still runs
  uint32_t* tail_flag = nullptr;
  uint32_t tail_gen = 0;
};

// what the device phase actually ran with (host memory, filled by msm_device_phase when the plan names one)
struct MsmTrace {
  int c = 0, nwin = 0, nb = 0, seg = 0, nseg = 0, run_serial = 0, w0 = 0, nwin_total = 0;
  const uint32_t* d_long_runs = nullptr;  // device: word 0 = number of runs that went to the work list
};

// restrict a whole plan to windows [w0, w0 + cnt): the sort chunking is re-derived for the smaller window count
inline void msm_plan_take_windows(MsmPlan& pl, int w0, int cnt, int q_blocks = 512) {
  pl.nwin_total = pl.nwin_total ? pl.nwin_total : pl.nwin;
  pl.w0 += w0;
  pl.nwin = cnt;
  int Q = cnt > 0 ? (q_blocks / cnt > 1 ? q_blocks / cnt : 1) : 1;
  const int cap = pl.n / 4096 > 1 ? pl.n / 4096 : 1;
  if (Q > cap) Q = cap;
  pl.Q = Q;
  pl.chunk = (pl.n + Q - 1) / Q;
  if (cnt < 8) {  // the XCD-local sort needs a window per XCD (msm_plan.hpp msm_plan_sort_locality)
    pl.xcd_map = 0;
    pl.scatter_passes = 1;
  }
}

// the plan the kernels AFTER the sort see: the plan itself, or - shared-bucket mode - one window holding every entry
inline MsmPlan msm_acc_view(const MsmPlan& pl) {
  MsmPlan av = pl;
  if (pl.shared) {
    av.n = pl.n * pl.nwin;
    av.nwin = 1;
  }
  return av;
}

// Group policy of the MSM kernels: how an input point is stored, what the bucket accumulator
// is and the three operations on it.  Default: short-Weierstrass a = 0 (XYZZ buckets, affine
// inputs); CurveEd specialises it for twisted Edwards (extended buckets, Niels inputs).
template <class C>
struct MsmGroup {
  using F = typename C::F;
  using Acc = Xyzz<F>;
  using Aff = Affine<F>;
  static constexpr int FW = FieldIO<F>::WORDS;
  static constexpr int WIRE_AFF = 2 * FieldWire<F>::WORDS;  // wire words per input point
  static constexpr int AFF_WORDS = 2 * FW;                  // stored words per input point
  static constexpr int ACC_WORDS = 4 * FW;                  // stored words per accumulator
  static NCG_DI void wire_to_storage(const uint32_t* wire, uint32_t* out) {
    Affine<F> a = load_affine_wire<F>(wire);
    FieldIO<F>::store(out, a.x);
    FieldIO<F>::store(out + FW, a.y);
  }
  static NCG_DI Aff aff_load(const uint32_t* p) { return {FieldIO<F>::load(p), FieldIO<F>::load(p + FW)}; }
  static NCG_DI Acc identity() { return Xyzz<F>::inf(); }
  static NCG_DI Acc acc_load(const uint32_t* p) {  // all-zero words (memset) decode as infinity
    return {FieldIO<F>::load(p), FieldIO<F>::load(p + FW), FieldIO<F>::load(p + 2 * FW), FieldIO<F>::load(p + 3 * FW)};
  }
  static NCG_DI void acc_store(uint32_t* p, const Acc& a) {
    FieldIO<F>::store(p, a.X);
    FieldIO<F>::store(p + FW, a.Y);
    FieldIO<F>::store(p + 2 * FW, a.ZZ);
    FieldIO<F>::store(p + 3 * FW, a.ZZZ);
  }
  static NCG_DI Acc madd(const Acc& a, const Aff& q, bool neg) { return xyzz_madd(a, q, neg); }
  static NCG_DI Acc add(const Acc& a, const Acc& b) { return xyzz_add(a, b); }
  static NCG_DI Acc dbl(const Acc& a) { return xyzz_dbl(a); }
  // canonical affine wire output; infinity = (0, 0)
  static NCG_DI void to_affine_wire(const Acc& acc, uint32_t* out, uint8_t* out_inf) {
    bool inf = acc.is_inf();
    Affine<F> A{F::zero(), F::zero()};
    if (!inf) {  // x = X/ZZ, y = Y/ZZZ; one inversion of ZZ*ZZZ
      auto ti = f_inv(acc.ZZ * acc.ZZZ);
      auto zzi = ti * acc.ZZZ;
      auto zzzi = ti * acc.ZZ;
      A = {acc.X * zzi, acc.Y * zzzi};
    }
    store_affine_wire<F>(out, A);
    *out_inf = inf ? 1 : 0;
  }
};

// Twisted Edwards (ed25519): inputs stored in affine Niels form (y+x, y-x, 2dxy), buckets in
// extended coordinates.  The formulas are complete, so the identity (0,1) needs no special
// case; an all-zero stored accumulator (memset) decodes as the identity.
template <>
struct MsmGroup<CurveEd> {
  using F = FEd;
  using Acc = EdExt<F>;
  using Aff = EdNielsAff<F>;
  static constexpr int FW = FieldIO<F>::WORDS;
  static constexpr int WIRE_AFF = 16;
  static constexpr int AFF_WORDS = 3 * FW;
  static constexpr int ACC_WORDS = 4 * FW;
  static NCG_DI void wire_to_storage(const uint32_t* wire, uint32_t* out) {
    F x = FieldWire<F>::load(wire), y = FieldWire<F>::load(wire + 8);
    Aff q = ed_affine_to_niels(x, y, EdConsts::d2());
    FieldIO<F>::store(out, q.yplusx);
    FieldIO<F>::store(out + FW, q.yminusx);
    FieldIO<F>::store(out + 2 * FW, q.t2d);
  }
  static NCG_DI Aff aff_load(const uint32_t* p) {
    return {FieldIO<F>::load(p), FieldIO<F>::load(p + FW), FieldIO<F>::load(p + 2 * FW)};
  }
  static NCG_DI Acc identity() { return EdExt<F>::identity(); }
  static NCG_DI Acc acc_load(const uint32_t* p) {
    Acc a{FieldIO<F>::load(p), FieldIO<F>::load(p + FW), FieldIO<F>::load(p + 2 * FW), FieldIO<F>::load(p + 3 * FW)};
    if (a.Z.is_zero()) a = EdExt<F>::identity();  // literal zeros = a memset slot
    return a;
  }
  static NCG_DI void acc_store(uint32_t* p, const Acc& a) {
    FieldIO<F>::store(p, a.X);
    FieldIO<F>::store(p + FW, a.Y);
    FieldIO<F>::store(p + 2 * FW, a.Z);
    FieldIO<F>::store(p + 3 * FW, a.T);
  }
  static NCG_DI Acc madd(const Acc& a, const Aff& q, bool neg) { return ed_madd_niels(a, q, neg); }
  static NCG_DI Acc add(const Acc& a, const Acc& b) { return ed_add_niels(a, ed_to_niels(b, EdConsts::d2()), false); }
  static NCG_DI Acc dbl(const Acc& a) { return ed_dbl(a); }
  // affine (x, y) = (X/Z, Y/Z); the identity is (0, 1) on Edwards curves (edwards.ts:606)
  static NCG_DI void to_affine_wire(const Acc& acc, uint32_t* out, uint8_t* out_inf) {
    F zi = f_inv(acc.Z);
    F x = acc.X * zi, y = acc.Y * zi;
    FieldWire<F>::store(out, x);
    FieldWire<F>::store(out + 8, y);
    *out_inf = (f_eqz(x) && f_eq(y, F::one())) ? 1 : 0;
  }
};

}  // namespace ncg
