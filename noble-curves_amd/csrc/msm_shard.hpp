// Slot format of the point-sharded (multi-GPU) MSM: what one shard contributes to the exchange.
// Shared by comm.hip (RCCL all-gather and the host-staged exchange) and the CPU twin of the tests.
//
//   slot = FinHeader (16 bytes) + grouped window sums (msm_fin_words accumulators, device storage format)
//
// The slot size is FIXED per curve - the largest grouped-sum array any window plan of that curve produces - so
// every rank posts the same byte count to the all-gather whatever plan it derived (round 2 sized the slot from
// the rank's own plan: ragged shards straddling a power of two then posted different counts - ADVICE r02).  Plans
// that disagree are caught AFTER the gather by the header check, with an error instead of a hang.
#pragma once
#include <cstdio>

#include "msm_finish.hpp"
#include "msm_plan.hpp"

namespace ncg {

struct FinHeader {  // first 16 bytes of every shard's slot: the plans must agree
  uint32_t c, nwin, words, curve;
};

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

// returns -1 if every header matches the plan, else the first offending shard (message in `msg`)
inline int msm_shard_check(const FinHeader* hs, int nparts, int curve, const MsmPlan& pl, size_t fin_words, char* msg, size_t msg_len) {
  for (int r = 0; r < nparts; r++)
    if (hs[r].c != (uint32_t)pl.c || hs[r].nwin != (uint32_t)pl.nwin || hs[r].words != (uint32_t)fin_words ||
        hs[r].curve != (uint32_t)curve) {
      snprintf(msg, msg_len,
               "noble-gpu: msm_sharded: shard %d planned c=%u nwin=%u words=%u (curve %u), this rank c=%d nwin=%d words=%zu "
               "(curve %d) - all ranks must pass the same curve and n_max",
               r, hs[r].c, hs[r].nwin, hs[r].words, hs[r].curve, pl.c, pl.nwin, fin_words, curve);
      return r;
    }
  return -1;
}

}  // namespace ncg
