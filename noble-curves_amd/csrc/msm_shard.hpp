// Slot format of the multi-GPU MSM: what one rank contributes to the exchange, and the checks / assembly every rank
// runs on the gathered slots.  Shared by comm.hip (RCCL all-gather and the host-staged exchange) and the CPU twin of
// the tests (hosttest.hip), so the header rules and the window assembly under test ARE the shipped ones.
//
//   slot = FinHeader (32 bytes) + grouped window sums (accumulators in device storage format), zero padded
//
// Three ways to cut one MSM (pippenger, src/abstract/curve.ts:863-905) over G ranks:
//   SHARD_POINTS          every rank holds a slice of the points and runs ALL windows on it; the slots are added term by
//                         term (the MSM is a sum over points, curve.ts:890-894) - weak scaling, ragged shards
//   SHARD_WINDOWS         every rank holds ALL points (replicated or resident sets) and runs a contiguous range of the
//                         windows (windows are independent until the final double-and-add chain, curve.ts:886-902); the
//                         slots are CONCATENATED, then the usual Horner - strong scaling: sort, accumulate and the
//                         throughput part of the fold all divide by G
//   SHARD_WINDOWS_SHARED  the same on a precomputed set (shared-bucket mode, msm.hpp): a rank's windows add into one
//                         bucket set, its slot is one grouped sum array, the slots are added
//
// The slot size is FIXED per curve - the largest grouped-sum array any window plan of that curve produces - so every
// rank posts the same byte count to the all-gather whatever plan it derived.  Plans that disagree are
// caught AFTER the gather by the header check, with an error instead of a hang; a scalar outside the group order
// (validateMSMScalars, curve.ts:398-404) travels in the header too, so that EVERY rank fails the call.
#pragma once
#include <cstdio>
#include <cstring>

#include "msm_finish.hpp"
#include "msm_plan.hpp"

namespace ncg {

enum ShardMode : uint32_t { SHARD_POINTS = 0, SHARD_WINDOWS = 1, SHARD_WINDOWS_SHARED = 2 };

struct FinHeader {  // first 32 bytes of every rank's slot
  uint32_t c, nwin, words, curve;  // window bits, windows of the WHOLE plan, payload words of this slot, curve id
  uint32_t w0, wcnt, mode, bad;    // this rank's window range, ShardMode, smallest index of a scalar >= the group order (or ~0)
};
static_assert(sizeof(FinHeader) == 32, "slot header");
constexpr uint32_t SHARD_NO_BAD = 0xFFFFFFFFu;

inline size_t msm_shard_max_fin_words(int curve) {
  size_t best = 0;
  for (int c = 2; c <= 16; c++) {
    MsmPlan pl;
    if (msm_make_plan_impl(curve, 1 << 20, c, &pl) != 0) continue;
    best = std::max(best, (size_t)msm_ngroups(pl.c) * pl.nwin * msm_acc_words_inl(curve));
  }
  return best;
}
inline size_t msm_shard_slot_bytes(int curve) {
  const size_t fw = msm_shard_max_fin_words(curve);
  return fw ? ((sizeof(FinHeader) + fw * 4 + 255) & ~(size_t)255) : 0;
}

// contiguous window range of rank `part` of `nparts` (sizes differ by at most one; ranks beyond the window count get none)
inline void msm_shard_window_range(int nwin, int part, int nparts, int* w0, int* cnt) {
  const int base = nwin / nparts, rem = nwin % nparts;
  *w0 = part * base + std::min(part, rem);
  *cnt = base + (part < rem ? 1 : 0);
}

// returns -1 if every header matches the whole plan `pl` and the mode's layout, else the first offending rank (message in `msg`)
inline int msm_shard_check(const FinHeader* hs, int nparts, int curve, const MsmPlan& pl, uint32_t mode, char* msg, size_t msg_len) {
  const size_t xw = msm_acc_words_inl(curve);
  const uint32_t ng = (uint32_t)msm_ngroups(pl.c);
  uint32_t next_w = 0;
  for (int r = 0; r < nparts; r++) {
    const FinHeader& h = hs[r];
    const uint32_t want_cnt = mode == SHARD_POINTS ? (uint32_t)pl.nwin : h.wcnt;
    const uint32_t want_words = (uint32_t)(ng * (mode == SHARD_WINDOWS_SHARED ? 1u : want_cnt) * xw);
    bool ok = h.c == (uint32_t)pl.c && h.nwin == (uint32_t)pl.nwin && h.curve == (uint32_t)curve && h.mode == mode && h.words == want_words;
    if (ok && mode == SHARD_POINTS) ok = h.w0 == 0 && h.wcnt == (uint32_t)pl.nwin;
    if (ok && mode != SHARD_POINTS) {  // the ranges tile [0, nwin) in rank order
      ok = h.w0 == next_w && h.w0 + h.wcnt <= (uint32_t)pl.nwin;
      next_w = h.w0 + h.wcnt;
    }
    if (!ok) {
      snprintf(msg, msg_len,
               "noble-gpu: msm_sharded: shard %d planned c=%u nwin=%u words=%u windows=[%u,+%u) mode=%u (curve %u), this rank c=%d nwin=%d "
               "mode=%u (curve %d) - all ranks must pass the same curve and n_max",
               r, h.c, h.nwin, h.words, h.w0, h.wcnt, h.mode, h.curve, pl.c, pl.nwin, mode, curve);
      return r;
    }
  }
  if (mode != SHARD_POINTS && next_w != (uint32_t)pl.nwin) {
    snprintf(msg, msg_len, "noble-gpu: msm_sharded: the ranks' window ranges cover %u of %d windows - all ranks must pass the same curve and n_max",
             next_w, pl.nwin);
    return nparts - 1;
  }
  return -1;
}

// the scalar-range verdict of all ranks: -1 if every scalar was in range, else the first rank that saw one (its index in *idx)
inline int msm_shard_first_bad(const FinHeader* hs, int nparts, uint32_t* idx) {
  for (int r = 0; r < nparts; r++)
    if (hs[r].bad != SHARD_NO_BAD) {
      *idx = hs[r].bad;
      return r;
    }
  return -1;
}

// SHARD_WINDOWS: the slots' [ngroups][wcnt_r] arrays (host memory, `stride` bytes apart) -> one [ngroups][nwin] array
inline void msm_shard_assemble_windows(const uint8_t* slots, size_t stride, const FinHeader* hs, int nparts, int curve, const MsmPlan& pl,
                                       uint32_t* fin_out) {
  const size_t xw = msm_acc_words_inl(curve);
  const int ng = msm_ngroups(pl.c);
  for (int r = 0; r < nparts; r++) {
    const uint32_t* src = (const uint32_t*)(slots + stride * (size_t)r + sizeof(FinHeader));
    const int w0 = (int)hs[r].w0, cnt = (int)hs[r].wcnt;
    for (int j = 0; j < ng; j++)
      if (cnt) memcpy(fin_out + ((size_t)j * pl.nwin + w0) * xw, src + (size_t)j * cnt * xw, (size_t)cnt * xw * 4);
  }
}

}  // namespace ncg
