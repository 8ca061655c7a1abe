// The ONE configuration surface of libncg.so: every environment variable the library reads, with its meaning.
// (The reference has no configuration beyond constructor options; these exist for diagnosis and
// for the A/B tooling.)  `ncg::knob(name, default)` is the only place that calls getenv.
//
//   always honoured
//     NCG_TIMING=1        print the MSM host-finish time to stderr per call
//     NCG_NO_ENDO=1       resident bls12-381 sets never take the endomorphism MSM / GLV ladder (generic path)
//     NCG_NO_PRECOMP=1    resident sets ignore their precomputed shifted copies (per-window path)
//     NCG_MSM_C=<c>       force the MSM window width (2..16) instead of the measured table (tools/msm_csweep.py)
//     NCG_MSM_C_ENDO=<c>  the same for the endomorphism plan
//     NCG_NO_FINISH_THREADS=1   the host finish of the bls12-381 MSMs never uses its three helper threads (bls_host64.hpp FinishPool;
//                         they are created at the first MSM that can use them, sleep between MSMs, and spin for at most 0.6 ms per MSM)
//     NCG_LANE_QUEUES=<m> hardware queues of the asynchronous MSM lanes (comm.hip lane_init): 1 = streams of the top priority class, a
//                         queue pool nothing else in the process uses (default); 0 = plain streams; 2 = a full CU mask per lane stream
//   A/B builds only (-DNCG_AB_BUILD, tools/ab_*.sh, tools/msm_debug.py): ignored by the shipped library
//     NCG_MSM_SEG, NCG_MSM_QBLOCKS, NCG_MSM_XCD     accumulate segment length, sort chunk count, XCD-aware sort grid
//     NCG_MSM_RUN_SERIAL, NCG_MSM_COOP_LEVEL        fix-up serial threshold, cooperative level kernel on / off
//     NCG_MSM_MERGE_UNITS, NCG_MSM_TOTALS_SPLIT     cooperative fix-up units for per-window plans (1 / 0), column walk of the count arrays split over lanes
//     NCG_MSM_MERGE_TREE, NCG_MSM_SMALL_SORT, NCG_MSM_ACCUM_SPREAD, NCG_MSM_TAIL_ROUNDS   the small-plan path of round 6, piece by piece (0 = off; rounds: 1 / 2)
//     NCG_NTT_LDS_PCT                               dynamic LDS of the NTT pass in percent of what it needs (occupancy experiment)
//     NCG_MSM_HOST64                                host finish in 64-bit limbs on / off
//     NCG_MSM_HOST_PARTS                            parts of the host-pointer MSM (1..8)
//     NCG_MULVAR_HOST_EVEN                          1 = four equal chunks in the host-pointer batch multiply instead of 1 : 3 : 3 : 1
//     NCG_SECP_W, NCG_G1_W, NCG_G2_W, NCG_AFF_K, NCG_ED_VARIANT, NCG_DEC_G2_FUSED, NCG_H2C_G2_FUSED   kernel variants
#pragma once
#include <cstdlib>
#include <cstring>

namespace ncg {

inline bool knob_is_public(const char* name) {
  static const char* const pub[] = {"NCG_TIMING", "NCG_NO_ENDO", "NCG_NO_PRECOMP", "NCG_MSM_C", "NCG_MSM_C_ENDO", "NCG_NO_FINISH_THREADS", "NCG_LANE_QUEUES"};
  for (const char* p : pub)
    if (std::strcmp(p, name) == 0) return true;
  return false;
}
// integer value of the variable, `dflt` when unset (or when the variable is A/B-only and this is not an A/B build)
inline int knob(const char* name, int dflt) {
#ifndef NCG_AB_BUILD
  if (!knob_is_public(name)) return dflt;
#endif
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}
inline bool knob_set(const char* name) {
#ifndef NCG_AB_BUILD
  if (!knob_is_public(name)) return false;
#endif
  return std::getenv(name) != nullptr;
}

}  // namespace ncg
