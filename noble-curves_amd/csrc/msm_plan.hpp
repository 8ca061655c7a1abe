// Window plan of the MSM (host side; shared by msm.hip, comm.hip and the CPU test twin).
#pragma once
#include <algorithm>
#include <cmath>
#include "knobs.hpp"
#include <cstdlib>

#include "msm.hpp"

// waves/SIMD of the lane-paired G2 accumulate kernel (the whole library must agree: the window plan sizes the lane segment
// for the kernel's resident-lane capacity).  2 since round 4: with the column-wise Montgomery product (fp29.hpp
// mont_cols29, NCG_FE29_COLS in msm.o) the kernel needs 232 registers instead of 306.
#ifndef NCG_G2_ACCUM_WAVES
#define NCG_G2_ACCUM_WAVES 2
#endif

namespace ncg {

inline void mp_set_bit(uint32_t* a, int bit) { a[bit >> 5] |= 1u << (bit & 31); }

// Largest scalar is order-1; choose the fewest windows with (order-1) + H' < 2^(c*nwin).
inline int plan_windows(int c, const uint32_t* order8, uint32_t* hconst10) {
  for (int nwin = (252 / c); nwin <= 300 / c + 2; nwin++) {
    if (nwin < 1 || c * nwin > 10 * 32 - 2) continue;
    uint32_t h[10] = {0};
    for (int w = 0; w < nwin; w++) mp_set_bit(h, c * w + c - 1);
    // s = (order - 1) + h
    uint32_t s[10];
    uint64_t cy = 0;
    for (int i = 0; i < 10; i++) {
      uint64_t o = i < 8 ? order8[i] : 0;
      uint64_t t = o + h[i] + cy;
      s[i] = (uint32_t)t;
      cy = t >> 32;
    }
    // subtract 1 (order >= 1): fine to skip - being conservative by one is harmless
    // check s < 2^(c*nwin)
    bool ok = (cy == 0);
    int top = c * nwin;
    for (int bit = top; ok && bit < 320; bit++)
      if (s[bit >> 5] & (1u << (bit & 31))) ok = false;
    if (ok) {
      for (int i = 0; i < 10; i++) hconst10[i] = h[i];
      return nwin;
    }
  }
  return -1;
}

inline const uint32_t* curve_order(int curve) {
  switch (curve) {
    case CURVE_SECP256K1: return Orders::SECP_N;
    case CURVE_ED25519: return Orders::ED_L;
    default: return Orders::BLS_R;
  }
}

inline int ilog2(unsigned x) {
  int r = 0;
  while (x >>= 1) r++;
  return r;
}

// Sort locality (round 4, measured with tools/exp_scatter.py, 2^20 G1 points: k_msm_scatter 220 -> 168 us, k_msm_hist 71 -> 50 us):
// window-major block ids put all blocks of a window on one XCD, and the scatter runs as four launches of a quarter of the
// buckets each, so that the region a launch writes (1 MB per window, two windows per XCD) stays in that XCD's 4 MB L2 and
// leaves it as whole lines.  Only with at least eight windows (fewer would leave XCDs idle: the window-sharded ranks).
inline void msm_plan_sort_locality(MsmPlan& pl) {
  static const int xcd = knob("NCG_MSM_XCD", -1);   // A/B builds: force on / off
  const bool on = xcd >= 0 ? xcd != 0 : pl.nwin >= 8;
  pl.xcd_map = on ? 1 : 0;
  pl.scatter_passes = on ? 4 : 1;
}

// Largest field the top window of a generic plan can hold: v = (k + H') >> c (nwin - 1) with k < order (conservative by one).
inline uint32_t msm_plan_top_vmax(const MsmPlan& pl) {
  const int nwt = pl.nwin_total ? pl.nwin_total : pl.nwin;
  uint32_t sum[11] = {0};
  uint64_t cy = 0;
  for (int i = 0; i < 10; i++) {
    const uint64_t t = (uint64_t)(i < 8 ? pl.order[i] : 0u) + pl.hconst[i] + cy;
    sum[i] = (uint32_t)t;
    cy = t >> 32;
  }
  sum[10] = (uint32_t)cy;
  const int bit = pl.c * (nwt - 1);
  const uint64_t two = ((uint64_t)sum[(bit >> 5) + 1] << 32) | sum[bit >> 5];
  return (uint32_t)(two >> (bit & 31)) & ((1u << pl.c) - 1u);
}
// MsmPlan::top_tb: spread a short top window over sub-buckets (its digits are v - half in [0, vmax - half]: H' carries the
// window's own half, so they are never negative) when it would use at most a quarter of the buckets.
inline void msm_plan_top_spread(MsmPlan& pl) {
  static const int on = knob("NCG_MSM_TOP_SPREAD", 1);   // A/B builds: 0 = off
  pl.top_tb = 0;
  pl.top_submask = 0;
  if (!on || pl.endo || pl.shared) return;
  const uint32_t half = 1u << (pl.c - 1), vmax = msm_plan_top_vmax(pl);
  const uint32_t maxd = vmax >= half ? vmax - half + 1u : half;   // largest digit (+ 1 of slack)
  int tb = 1;
  while ((1u << tb) < maxd) tb++;
  // digits are stored as int16: the spread digit (sub << tb) + d must stay below 2^15
  const uint32_t subs = (pl.nb < 32768 ? (uint32_t)pl.nb : 16384u) >> tb;
  // at least 8 sub-buckets per digit value, or the runs only get twice as many (measured: ed25519 at c = 16, 12-bit top window,
  // 2 sub-buckets: k_msm_fixup_long 138 -> 209 us; G2 at c = 13, 8 bits, 16 sub-buckets: 59 -> 5 us)
  if (subs < 8) return;
  pl.top_tb = tb;
  pl.top_submask = subs - 1u;
}

inline int msm_make_plan_impl(int curve, int n, int c_override, MsmPlan* pl) {
  int c = c_override;
  if (c <= 0) {
    c = knob("NCG_MSM_C", 0);
  }
  if (c <= 0) {
    // accumulate cost nwin*n mixed adds vs fold cost ~2*nwin*2^(c-1) full adds: c ~ log2(n) - 4 - but the width also
    // decides how many bits the TOP window holds: 255-bit scalars in windows of 12 or 14 bits leave it 3 bits, i.e. a
    // handful of buckets holding every point of the window, which the fix-up then has to merge as very long runs
    // (measured: G1 2^17 c = 12 2.7 ms, c = 13 1.06 ms).  For bls12-381 the widths below are the measured best per
    // size on MI355X (tools/msm_csweep.py): round 3 (profiles/r03_msm_csweep.json) 8, 9, 10, 13, 15, 16; re-measured in round 5
    // with the two-level sort and the spread top windows (profiles/r05_msm_csweep.json): a short top window no longer costs long
    // fix-up runs, so c = 10 wins up to 2^15 (G1 2^15 0.73 -> 0.61 ms, G2 2^15 1.41 -> 1.24 ms) and c = 16 from 2^19.  Round 6
    // (short lane segments, merge tree and one-launch sort for the plans that do not fill the chip; profiles/r06_msm_small_sweep_c.json):
    // G1 2^10 c = 7 (0.36 against 0.41 ms at 6), 2^11 c = 8 (0.37 / 0.41), 2^16 c = 10 (0.76 / 0.87 at 13); G2 2^11 c = 8 (0.57 / 0.63).
    const int lg = ilog2((unsigned)std::max(n, 1));
    if (curve == CURVE_BLS12_381_G1) {
      static const int8_t tab[21] = {2, 2, 2, 2, 2, 2, 2, 3, 4, 5, 7, 8, 9, 10, 10, 10, 10, 13, 15, 16, 16};
      c = tab[std::min(lg, 20)];
    } else if (curve == CURVE_BLS12_381_G2) {
      static const int8_t tab[21] = {2, 2, 2, 2, 2, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10, 13, 13, 13, 15, 16};
      c = tab[std::min(lg, 20)];
    } else {
      c = lg - 4;
    }
  }
  c = std::max(2, std::min(16, c));
  pl->n = n;
  pl->ls = curve == CURVE_BLS12_381_G2 ? 1 : 0;  // lane-paired kernels: 2 lanes per item
  // waves/SIMD the accumulate kernel runs at (registers): 4 for the 256-bit fields, 2 for bls12-381 G1, 1 for G2
  pl->accum_waves = (curve == CURVE_SECP256K1 || curve == CURVE_ED25519) ? 4 : curve == CURVE_BLS12_381_G2 ? NCG_G2_ACCUM_WAVES : 2;
  pl->c = c;
  pl->nb = 1 << (c - 1);
  pl->nwin = plan_windows(c, curve_order(curve), pl->hconst);
  if (pl->nwin < 0) return -1;
  for (int i = 0; i < 8; i++) pl->order[i] = curve_order(curve)[i];
  // sort chunks: ~512 blocks per sort kernel (two per CU; measured 2 % faster than 1024 on the 2^20 G1 MSM,
  // tools/ab_q.sh: half the per-chunk count arrays to write, prefix and read), at least 4096 points per chunk
  static const int q_blocks = std::max(64, knob("NCG_MSM_QBLOCKS", 512));
  int Q = std::max(1, q_blocks / pl->nwin);
  Q = std::min(Q, std::max(1, n / 4096));
  pl->Q = Q;
  pl->chunk = (n + Q - 1) / Q;
  msm_plan_sort_locality(*pl);
  msm_plan_top_spread(*pl);
  return 0;
}



// ---- the lane segment of the accumulate kernel (k_msm_accum: every lane adds exactly `seg` consecutive entries of a window's sorted list)
struct MsmSeg {
  int seg;    // entries per lane
  int nseg;   // lanes per window = ceil(n / seg)
};
inline MsmSeg msm_seg(const MsmPlan& pl) {
  MsmSeg sg;
  const int seg_knob = pl.seg_override > 0 ? pl.seg_override : knob("NCG_MSM_SEG", 0);
  if (seg_knob > 0) {
    sg.seg = seg_knob;
  } else {
    // Every lane adds `seg` consecutive sorted entries, and the accumulate kernel keeps
    // cap = waves/SIMD x 1024 SIMDs x 64 lanes resident, so its time goes like rounds(seg) * seg with
    // rounds = ceil(lanes / cap), plus ~3 addition-times of fix-up per lane.  Pick the seg that minimises
    // that (measured on MI355X: G1 2^20 seg 128 = one full round 4.29 ms, 64 = two rounds 4.46, 96 4.90,
    // 192 5.44; G2 2^18 seg 73 = one round 4.84 ms, 32 5.08, 74 5.14).
    const long cap = 65536L * pl.accum_waves;
    double best = 1e300;
    sg.seg = 16;
    const int n_seg = pl.n_layout > 0 ? pl.n_layout : pl.n;   // parts of one MSM share the segment of the layout plan
    // Plans that do NOT fill the chip (round 6, tools/msm_small_sweep.py -> profiles/r06_msm_small_sweep.json): the kernel's time is
    // then seg x the latency of one addition on a SIMD that holds k = 1, 2.. waves - max(r1, k) in units of the throughput-bound
    // time of a wave-addition, r1 = a lone wave's latency (G1 12 us against 9.5, lane-paired G2 16 against 14.6) - and shorter
    // segments cut every bucket of m = n / nb entries into m / seg more pieces, which its fix-up unit adds one after the other
    // (f = one cooperative addition, 8 / 12 us; single-lane complete additions on the curves without cooperative units).  With
    // the floor of 16 entries the G1 plans below 2^15 points ran 350-420 waves on 1024 SIMDs for 16 x 12 us; measured best
    // segments: 4-6 up to 2^12 points, 6-8 at 2^13 / 2^14, 12 at 2^15 / 2^16 (G1 2^13 0.536 -> 0.466 ms, 2^16 0.85 -> 0.75 ms).
    const bool bls = pl.accum_waves == 2;
    const double r1 = pl.ls ? 1.1 : 1.25, f = bls ? 0.8 : 1.6;
    const double m = (double)n_seg / (double)std::max(1, pl.nb);
    for (int seg = 4; seg <= 160; seg++) {
      const long nseg = (n_seg + seg - 1) / seg;
      const long lanes = ((long)pl.nwin * nseg) << pl.ls;
      const long rounds = (lanes + cap - 1) / cap;
      double cost;
      if (lanes <= cap) {
        // waves per SIMD: the accumulate launch pins grids of up to 256 / 512 workgroups to one / two per CU (LDS reservation)
        const long wgs = (long)pl.nwin * (((nseg << pl.ls) + 255) / 256);
        const double k = wgs <= 256 ? 1.0 : wgs <= 512 ? 2.0 : (double)((lanes + 65535) / 65536);
        // pieces of the fullest buckets: (m + 4 sqrt(m)) / seg + 1; the merge adds them as a tree of four units per bucket where the
        // curve has cooperative units (k_msm_fixup_merge_tree), one after the other elsewhere
        const double pieces = (m + 4.0 * std::sqrt(m)) / (double)seg + 1.0;
        const double chain = bls ? std::max(0.0, std::ceil(pieces / 4.0) - 1.0) + 2.0 : pieces - 1.0;
        cost = (double)seg * std::max(r1, k) + std::max(3.0 * (double)lanes / 65536.0, f * chain);
      } else {
        if (seg < 16) continue;   // full chip: the measured segments of rounds 3-5 (16..160) stand
        cost = (double)rounds * seg * pl.accum_waves + 3.0 * (double)lanes / 65536.0;
      }
      if (cost < best - 1e-9) {
        best = cost;
        sg.seg = seg;
      }
    }
  }
  sg.nseg = (pl.n + sg.seg - 1) / sg.seg;
  return sg;
}


inline size_t msm_acc_words_inl(int curve) {
  switch (curve) {
    case CURVE_SECP256K1: return MsmGroup<CurveSecp>::ACC_WORDS;
    case CURVE_BLS12_381_G1: return MsmGroup<CurveG1>::ACC_WORDS;
    case CURVE_BLS12_381_G2: return MsmGroup<CurveG2>::ACC_WORDS;
    case CURVE_ED25519: return MsmGroup<CurveEd>::ACC_WORDS;
    default: return 0;
  }
}

}  // namespace ncg
