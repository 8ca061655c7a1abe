// Internal: the context behind the opaque ncg_ctx of include/ncg.h, shared by the translation units
// that implement the C ABI (api.hip, comm.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ncg.h"
#include "host_api.hpp"
#include "msm.hpp"

// One MSM in flight of the asynchronous entry points (ncg_msm_async_submit / _collect): its own stream, workspace,
// gather buffer and pinned landing area, so that the dependent tail of one MSM (narrow fold levels, per-window tail,
// D2H, host Horner) overlaps the sort / accumulate kernels of the next.
constexpr int NCG_MSM_LANES = 4;
struct ncg_msm_lane {
  hipStream_t stream = nullptr;
  ncg::MsmSide side;
  hipEvent_t done = nullptr, input_ready = nullptr;
  void* ws = nullptr;
  size_t ws_bytes = 0;
  uint32_t* land = nullptr;  // pinned host memory
  size_t land_words = 0;
  void* comm_buf = nullptr;  // window-sharded mode: the slots of all ranks (device)
  size_t comm_buf_bytes = 0;
  bool busy = false, state_identity = false, part_only = false;
  int curve = 0, mode = 0, nparts = 1, c = 0, nwin = 0;  // what collect needs to know about the enqueued job
  size_t stride = 0;
};

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
  int msm_seg_override = 0, msm_run_serial_override = -1;  // ncg_msm_set_tuning
  ncg::MsmTrace msm_trace;                                  // what the last MSM launch used (ncg_msm_last_plan)
  ncg_msm_lane lanes[NCG_MSM_LANES];                        // asynchronous MSMs in flight
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
  // host-pointer entry points (ncg_msm, ncg_mul_var_batch): copy streams and per-chunk events, so that the chunks of a
  // large host buffer cross PCIe while the kernels of earlier chunks (or the digit / sort kernels of the MSM) run
  static constexpr int COPY_CHUNKS = 8;
  hipStream_t copy_in = nullptr, copy_out = nullptr;
  hipEvent_t ev_in[COPY_CHUNKS] = {}, ev_k[COPY_CHUNKS] = {}, ev_sc[COPY_CHUNKS] = {}, ev_ready = nullptr;
  hipStream_t comm_stream = nullptr;  // the one stream every collective of the asynchronous lanes is enqueued on
  hipEvent_t comm_fork = nullptr, comm_join = nullptr;
  uint32_t* sync_land = nullptr;  // pinned landing area of the synchronous sharded entry points
  size_t sync_land_words = 0;
};


// ---- resident point sets (api.hip): upload once, multiply many (interleavedMSMUnsafe's usage pattern,
// src/abstract/curve.ts:907-959; SURVEY 8a gotcha 8: marshalling dominates an end-to-end call)
struct ncg_points {
  ncg_ctx* ctx;
  int curve;
  size_t n;
  void* d_pts;
  void* d_endo = nullptr;  // endomorphism images (msm_endo_expand) once the set is known to lie in the subgroup
  void* d_stored = nullptr;  // the points in the accumulate kernel's storage format (built at the first generic MSM)
  // window-shifted copies for the shared-bucket MSM (ncg_points_precompute, msm_precomp.hip): shift_nwin levels of
  // shift_m stored points; shift_mode 1 = levels of the points themselves, 2 = of the endomorphism images
  void* d_shift = nullptr;
  int shift_c = 0, shift_nwin = 0, shift_mode = 0;
  size_t shift_m = 0;
};

// which window plan and which device point array an MSM on a resident set uses (api.hip)
int ncg_resident_plan(ncg_ctx* ctx, const ncg_points* pts, ncg::MsmPlan* pl, const uint32_t** d_pts, hipStream_t st);
// whole plan for n points restricted to windows [w0, w0 + cnt), workspace grown as needed (ws / ws_bytes NULL: the context's own)
int ncg_msm_plan_ws_windows(ncg_ctx* ctx, int curve, size_t n, int w0, int cnt, ncg::MsmPlan* pl, void** ws, size_t* ws_bytes);
// tuning overrides + trace slot of the context into a plan, workspace (any of the context's) grown as needed
int ncg_msm_ensure_buf(ncg_ctx* ctx, int curve, ncg::MsmPlan& pl, void** ws, size_t* ws_bytes);

// records the message (per context and globally) and returns `code`
int ncg_set_err(ncg_ctx* ctx, int code, const char* fmt, ...);
#define set_err ncg_set_err

int ncg_msm_plan_ws(ncg_ctx* ctx, int curve, size_t n, int c_override, ncg::MsmPlan* pl);

#define NCG_HIP(ctx, expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return set_err(ctx, NCG_ERR_HIP, "noble-gpu: HIP error %d (%s) at %s:%d", (int)_e,    \
                     hipGetErrorString(_e), __FILE__, __LINE__);                             \
  } while (0)

