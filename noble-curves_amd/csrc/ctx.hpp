// Internal: the context behind the opaque ncg_ctx of include/ncg.h, shared by the translation units
// that implement the C ABI (api.hip, comm.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ncg.h"
#include "host_api.hpp"

struct ncg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string last_error;
  // reusable device scratch for the host-pointer entry points
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  void* mul_ws = nullptr;  // batch-multiply Jacobian scratch (device)
  size_t mul_ws_bytes = 0;
  void* msm_ws = nullptr;  // MSM workspace (device)
  size_t msm_ws_bytes = 0;
  ncg::MsmSide msm_side;   // second stream + fork / join events of the MSM (msm.hip)
  uint32_t* ed_btab = nullptr;  // ed25519 base-point table (device)
  void* ed_ks = nullptr;        // ed25519 challenge scalars of the message-taking verify (device)
  size_t ed_ks_bytes = 0;
  void* ecdsa_ws = nullptr;     // ECDSA batch verify: decoded keys, u1 / u2, partial points (device)
  size_t ecdsa_ws_bytes = 0;
  uint32_t* base_tab[4] = {nullptr, nullptr, nullptr, nullptr};  // fixed-base tables per curve (device)
  uint32_t* ub_in = nullptr;
  uint32_t* ub_out = nullptr;
  size_t ub_out_words = 0;
  // NTT: one twiddle table per transform size (device), keyed by the root it was built from
  uint32_t* ntt_tab[NCG_NTT_MAX_LOG2N + 1] = {};
  uint32_t ntt_omega[NCG_NTT_MAX_LOG2N + 1][8] = {};
  void* ntt_ws = nullptr;
  size_t ntt_ws_bytes = 0;
  // multi-GPU (comm.hip): RCCL communicator of this rank and the gather buffer of the sharded MSM
  void* comm = nullptr;  // ncclComm_t
  int comm_rank = 0, comm_size = 1;
  void* comm_buf = nullptr;
  size_t comm_buf_bytes = 0;
};


// records the message (per context and globally) and returns `code`
int ncg_set_err(ncg_ctx* ctx, int code, const char* fmt, ...);
#define set_err ncg_set_err

namespace ncg { struct MsmPlan; }
int ncg_msm_plan_ws(ncg_ctx* ctx, int curve, size_t n, int c_override, ncg::MsmPlan* pl);

#define NCG_HIP(ctx, expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return set_err(ctx, NCG_ERR_HIP, "noble-gpu: HIP error %d (%s) at %s:%d", (int)_e,    \
                     hipGetErrorString(_e), __FILE__, __LINE__);                             \
  } while (0)

