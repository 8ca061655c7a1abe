// Multi-GPU MSM behind the C ABI (include/ncg.h, "multi-GPU" section).
//
// pippenger is linear in its points (src/abstract/curve.ts:863-905: every point lands in one bucket per
// window and the result is a sum over points), so the points are sharded: every GPU runs the whole
// single-GPU pipeline on its slice up to the grouped window sums (msm.hpp step 5b: ng x nwin
// accumulators, ~18 KB for G1), ONE ncclAllGather moves those over xGMI, a one-wave kernel adds the G
// arrays element by element (G - 1 additions per lane) and the usual finish runs on the sum.  Bucket-
// sized data never moves; RCCL cannot reduce with a group law, so the "all-reduce" of the north star
// is all-gather + local add.
//
// Two shapes, same kernels:
//   * one process per GPU (bench.py / torch.distributed / MPI): ncg_comm_unique_id + ncg_comm_init on a
//     context, then ncg_msm_sharded_dev - a collective: every rank calls it, every rank gets the result.
//   * one process, several GPUs (the N-API addon: Node is single-threaded): ncg_multi_init opens one
//     context per device and ncclCommInitAll; ncg_msm_multi shards host arrays over them.
// RCCL is loaded with dlopen on first use, so single-GPU users never touch it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <atomic>
#include <vector>

#include "ctx.hpp"
#include "host_api.hpp"
#include "msm.hpp"
#include "msm_shard.hpp"

using ncg::FinHeader;

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;        // optional (diagnostics: ncg_comm_count)
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl* rccl() {
  std::call_once(g_rccl_once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return;
#define NCG_SYM(field, sym) g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.lib, #sym)
    NCG_SYM(GetUniqueId, ncclGetUniqueId);
    NCG_SYM(CommInitRank, ncclCommInitRank);
    NCG_SYM(CommInitAll, ncclCommInitAll);
    NCG_SYM(CommDestroy, ncclCommDestroy);
    NCG_SYM(CommCount, ncclCommCount);
    NCG_SYM(CommUserRank, ncclCommUserRank);
    NCG_SYM(AllGather, ncclAllGather);
    NCG_SYM(GroupStart, ncclGroupStart);
    NCG_SYM(GroupEnd, ncclGroupEnd);
    NCG_SYM(GetErrorString, ncclGetErrorString);
#undef NCG_SYM
  });
  const Rccl& r = g_rccl;
  if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart ||
      !r.GroupEnd || !r.GetErrorString)
    return nullptr;
  return &g_rccl;
}

#define NCG_NCCL(ctx, r, expr)                                                                                \
  do {                                                                                                        \
    ncclResult_t _n = (expr);                                                                                 \
    if (_n != ncclSuccess)                                                                                    \
      return set_err(ctx, NCG_ERR_RCCL, "noble-gpu: RCCL error %d (%s) at %s:%d", (int)_n, (r)->GetErrorString(_n), \
                     __FILE__, __LINE__);                                                                     \
  } while (0)

// like NCG_HIP, for code that has asynchronous copies into host objects in flight on `st`: drain the stream
// before the early return destroys them
#define NCG_HIP_DRAIN(ctx, st, expr)                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      (void)hipStreamSynchronize(st);                                                         \
      return set_err(ctx, NCG_ERR_HIP, "noble-gpu: HIP error %d (%s) at %s:%d", (int)_e,     \
                     hipGetErrorString(_e), __FILE__, __LINE__);                              \
    }                                                                                         \
  } while (0)

// ---- one sharded MSM job on one rank ----------------------------------------------------------------------------
// The resources a job runs on: the context's own (synchronous entry points, caller's stream) or one of its lanes
// (asynchronous entry points: several MSMs in flight).
struct JobRes {
  hipStream_t st;
  const ncg::MsmSide* side;
  void** ws;
  size_t* ws_bytes;
  void** comm_buf;       // device: nparts slots, then the packed payloads + their sum (point / shared modes)
  size_t* comm_buf_bytes;
  uint32_t** land;       // pinned host: the gathered slots, then the summed payload
  size_t* land_words;
  bool funnel = false;   // asynchronous lanes: the collective goes through the context's one communication stream
};
// what finish needs to know about an enqueued job
struct JobState {
  int curve = 0, nparts = 1;
  uint32_t mode = ncg::SHARD_POINTS;
  int c = 0, nwin = 0;   // the WHOLE plan
  size_t stride = 0, sum_words = 0;
  bool identity = false;  // nothing to do: the result is the identity (curve.ts:878)
  // "the tail kernel of this rank's local phase has started" (a word of the pinned landing area, set by k_msm_tail to tail_gen):
  // the synchronous entry points poll it and wake the helper threads of the host finish while the tail, the exchange and the
  // combine still run (wait_tail_then_wake)
  volatile uint32_t* tail_flag = nullptr;
  uint32_t tail_gen = 0;
  size_t slice = 0;       // SHARD_POINTS from ONE caller (ncg_msm_split_dev, ncg_msm_multi): points per part, so that a bad scalar
                          // is reported with its index in the caller's arrays ('invalid scalar at index i', curve.ts:402); 0 = the
                          // caller is one rank of several and only knows shard-relative indices
};
struct ShardJob {
  int curve = 0;
  uint32_t mode = ncg::SHARD_POINTS;
  size_t n_local = 0, n_plan = 0;      // SHARD_POINTS: this rank's slice and the largest slice of any rank; window modes: n both
  const void* d_pts = nullptr;         // device wire points (or NULL with `resident`)
  const ncg_points* resident = nullptr;
  const void* d_sc = nullptr;
  int part = 0, nparts = 1;            // this rank's place in the exchange
};

int ensure_dev_buf(ncg_ctx* ctx, void** p, size_t* have, size_t bytes) {
  if (*have >= bytes) return NCG_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *have = 0;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  *have = bytes;
  return NCG_OK;
}
int ensure_land(ncg_ctx* ctx, uint32_t** p, size_t* have_words, size_t words) {
  if (*have_words >= words) return NCG_OK;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *have_words = 0;
  hipError_t e = hipHostMalloc((void**)p, words * 4, hipHostMallocPortable);
  if (e != hipSuccess) return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipHostMalloc(%zu) failed: %s", words * 4, hipGetErrorString(e));
  *have_words = words;
  return NCG_OK;
}
size_t comm_buf_need(size_t stride, size_t fin_words, int nparts) {
  return stride * (size_t)nparts + ((size_t)nparts + 1) * fin_words * 4 + 256;
}

// This rank's part: header + grouped window sums into slot `slot_idx` of the gather buffer.  Returns the whole plan.
// Window plan shared by all ranks of one MSM: every rank must cut the same windows, so in SHARD_POINTS the width is chosen
// for the LARGEST slice (the caller passes it) - which also re-tunes c for the slice size: the bucket fold costs ~2^c per
// window whatever the slice holds; in the window modes every rank holds all n points and derives the same plan.
int job_local_phase(ncg_ctx* ctx, const JobRes& R, const ShardJob& J, int slot_idx, int nslots, JobState* S) {
  const int curve = J.curve;
  ncg::MsmPlan whole, local;
  const uint32_t* d_pts = (const uint32_t*)J.d_pts;
  uint32_t mode = J.mode;
  int rc;
  if (J.resident) {
    rc = ncg_resident_plan(ctx, J.resident, &whole, &d_pts, R.st);
    if (rc) return rc;
    if (whole.shared && mode == ncg::SHARD_WINDOWS) mode = ncg::SHARD_WINDOWS_SHARED;
  } else {
    if (ncg::msm_make_plan(curve, (int)J.n_plan, 0, &whole) != 0) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: cannot plan windows");
  }
  local = whole;
  int w0 = 0, cnt = whole.nwin;
  if (mode == ncg::SHARD_POINTS) {
    if (J.n_local != J.n_plan && J.n_local > 0) {  // same windows, this slice's size
      if (ncg::msm_make_plan(curve, (int)J.n_local, whole.c, &local) != 0) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: cannot plan windows");
    }
  } else {
    ncg::msm_shard_window_range(whole.nwin, J.part, J.nparts, &w0, &cnt);
    // one or two windows per rank: 64 sort chunks per window instead of 256 (k_msm_bucket_totals walks the chunks serially:
    // 64 -> 21 us of a 0.9 ms share; measured 0.905 -> 0.856 ms for G = 8, no difference from 4 windows per rank up)
    static const int share_q = ncg::knob("NCG_MSM_SHARE_QBLOCKS", 128);   // A/B builds
    ncg::msm_plan_take_windows(local, w0, cnt, cnt <= 2 ? share_q : 512);
  }
  const size_t xw = ncg::msm_acc_words(curve);
  const size_t ng = (size_t)ncg::msm_ngroups(whole.c);
  const size_t fin_words = ng * (mode == ncg::SHARD_WINDOWS_SHARED ? 1 : (size_t)cnt) * xw;
  const size_t stride = ncg::msm_shard_slot_bytes(curve);  // the same on every rank, whatever its plan
  rc = ensure_dev_buf(ctx, R.comm_buf, R.comm_buf_bytes, comm_buf_need(stride, ncg::msm_shard_max_fin_words(curve), nslots));
  if (rc) return rc;
  const bool work = J.n_local > 0 && cnt > 0;
  if (work) {
    rc = ncg_msm_ensure_buf(ctx, curve, local, R.ws, R.ws_bytes);
    if (rc) return rc;
  }
  char* mine = (char*)*R.comm_buf + stride * (size_t)slot_idx;
  FinHeader h{(uint32_t)whole.c, (uint32_t)whole.nwin, (uint32_t)fin_words, (uint32_t)curve, (uint32_t)w0, (uint32_t)cnt, mode, ncg::SHARD_NO_BAD};
  // (the header is read by the copy engine before this function's frame dies only if the source is pinned: stage it in `land`)
  static_assert(sizeof(FinHeader) % 4 == 0, "header words");
  rc = ensure_land(ctx, R.land, R.land_words, (stride * (size_t)nslots + ncg::msm_shard_max_fin_words(curve) * 4) / 4 + 64);
  if (rc) return rc;
  FinHeader* hstage = (FinHeader*)(*R.land) + slot_idx;  // first bytes of the landing area; overwritten by the gather's D2H later
  *hstage = h;
  NCG_HIP(ctx, hipMemcpyAsync(mine, hstage, sizeof h, hipMemcpyHostToDevice, R.st));
  S->tail_flag = nullptr;
  if (!work) {  // an empty slice / no windows: identities (all-zero accumulators decode as such)
    if (fin_words) NCG_HIP(ctx, hipMemsetAsync(mine + sizeof h, 0, fin_words * 4, R.st));
  } else {
    {  // the last word of the landing area (64 words of slack behind everything the copies write)
      static std::atomic<uint32_t> gen{0};
      uint32_t g = gen.fetch_add(1, std::memory_order_relaxed) + 1;
      if (g == 0) g = gen.fetch_add(1, std::memory_order_relaxed) + 1;
      uint32_t* flag = *R.land + (*R.land_words - 1);
      *flag = 0;
      local.tail_flag = flag;
      local.tail_gen = g;
      S->tail_flag = flag;
      S->tail_gen = g;
    }
    const uint32_t *d_fin = nullptr, *d_bad = nullptr;
    NCG_HIP(ctx, ncg::msm_device_phase(curve, local, d_pts, (const uint32_t*)J.d_sc, *R.ws, &d_fin, R.st, &d_bad,
                                       (local.pts_stored || local.endo) ? nullptr : R.side));
    NCG_HIP(ctx, hipMemcpyAsync(mine + sizeof h, d_fin, fin_words * 4, hipMemcpyDeviceToDevice, R.st));
    NCG_HIP(ctx, hipMemcpyAsync(mine + offsetof(FinHeader, bad), d_bad, 4, hipMemcpyDeviceToDevice, R.st));  // the verdict travels with the slot
  }
  S->curve = curve;
  S->mode = mode;
  S->c = whole.c;
  S->nwin = whole.nwin;
  S->stride = stride;
  return NCG_OK;
}

// After the gather (or with all slots written locally): what can be enqueued without looking at the data - the slots
// travel to the host; in the adding modes the payloads are packed, summed on the device (pairwise tree, cooperative
// additions) and the sum follows the slots.
int job_enqueue_combine(ncg_ctx* ctx, const JobRes& R, JobState* S, int nparts) {
  S->nparts = nparts;
  const int curve = S->curve;
  char* base = (char*)*R.comm_buf;
  const size_t stride = S->stride;
  uint32_t* land = *R.land;
  NCG_HIP_DRAIN(ctx, R.st, hipMemcpyAsync(land, base, stride * (size_t)nparts, hipMemcpyDeviceToHost, R.st));
  S->sum_words = 0;
  if (S->mode != ncg::SHARD_WINDOWS) {
    const size_t xw = ncg::msm_acc_words(curve);
    const size_t fin_words = (size_t)ncg::msm_ngroups(S->c) * (S->mode == ncg::SHARD_WINDOWS_SHARED ? 1 : (size_t)S->nwin) * xw;
    const uint32_t* sum = (const uint32_t*)(base + sizeof(FinHeader));
    if (nparts > 1) {  // pack payloads: [nparts][fin_words] behind the slots (sized by comm_buf_need), reduced in place by the adding kernel
      uint32_t* packed = (uint32_t*)(base + stride * (size_t)nparts);
      for (int r = 0; r < nparts; r++)
        NCG_HIP_DRAIN(ctx, R.st, hipMemcpyAsync(packed + (size_t)r * fin_words, base + stride * (size_t)r + sizeof(FinHeader), fin_words * 4,
                                                hipMemcpyDeviceToDevice, R.st));
      uint32_t* dsum = packed + (size_t)nparts * fin_words;
      NCG_HIP_DRAIN(ctx, R.st, ncg::msm_sum_partials(curve, packed, nparts, fin_words / xw, dsum, R.st));
      sum = dsum;
    }
    NCG_HIP_DRAIN(ctx, R.st, hipMemcpyAsync(land + stride * (size_t)nparts / 4, sum, fin_words * 4, hipMemcpyDeviceToHost, R.st));
    S->sum_words = fin_words;
  }
  return NCG_OK;
}

// With the stream drained: header check, scalar verdict of all ranks, window assembly, host Horner.
int job_finish_host(ncg_ctx* ctx, const JobRes& R, const JobState& S, void* out_affine, uint8_t* out_is_inf) {
  const int nparts = S.nparts, curve = S.curve;
  const uint8_t* slots = (const uint8_t*)*R.land;
  std::vector<FinHeader> hs(nparts);
  for (int r = 0; r < nparts; r++) memcpy(&hs[r], slots + S.stride * (size_t)r, sizeof(FinHeader));
  ncg::MsmPlan pl;  // only c and nwin matter from here on
  pl.c = S.c;
  pl.nwin = S.nwin;
  char msg[400];
  if (ncg::msm_shard_check(hs.data(), nparts, curve, pl, S.mode, msg, sizeof msg) >= 0) return set_err(ctx, NCG_ERR_INVALID_ARG, "%s", msg);
  uint32_t bad_idx = 0;
  const int bad_rank = ncg::msm_shard_first_bad(hs.data(), nparts, &bad_idx);
  if (bad_rank >= 0) {
    if (S.mode == ncg::SHARD_POINTS && nparts > 1 && S.slice)
      return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: invalid scalar at index %zu (not below the group order)", S.slice * (size_t)bad_rank + bad_idx);
    if (S.mode == ncg::SHARD_POINTS && nparts > 1)
      return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_sharded: invalid scalar at index %u of shard %d (not below the group order)", bad_idx, bad_rank);
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: invalid scalar at index %u (not below the group order)", bad_idx);
  }
  uint8_t inf_local = 0;
  if (S.mode == ncg::SHARD_WINDOWS) {
    std::vector<uint32_t> fin((size_t)ncg::msm_ngroups(S.c) * S.nwin * ncg::msm_acc_words(curve));
    ncg::msm_shard_assemble_windows(slots, S.stride, hs.data(), nparts, curve, pl, fin.data());
    ncg::msm_finish_host(curve, S.c, S.nwin, fin.data(), (uint32_t*)out_affine, &inf_local);
  } else {
    const uint32_t* sum = *R.land + S.stride * (size_t)nparts / 4;
    ncg::msm_finish_host(curve, S.c, S.mode == ncg::SHARD_WINDOWS_SHARED ? 1 : S.nwin, sum, (uint32_t*)out_affine, &inf_local);
  }
  if (out_is_inf) *out_is_inf = inf_local;
  return NCG_OK;
}

JobRes ctx_res(ncg_ctx* ctx, hipStream_t st) {
  return JobRes{st, &ctx->msm_side, &ctx->msm_ws, &ctx->msm_ws_bytes, &ctx->comm_buf, &ctx->comm_buf_bytes, &ctx->sync_land, &ctx->sync_land_words};
}
JobRes lane_res(ncg_msm_lane& ln) {
  return JobRes{ln.stream, &ln.side, &ln.ws, &ln.ws_bytes, &ln.comm_buf, &ln.comm_buf_bytes, &ln.land, &ln.land_words, true};
}

int identity_out(int curve, void* out_affine, uint8_t* out_is_inf) {
  const int pb = ncg_point_bytes(curve);
  memset(out_affine, 0, pb);
  if (curve == NCG_ED25519) ((uint8_t*)out_affine)[32] = 1;
  if (out_is_inf) *out_is_inf = 1;
  return NCG_OK;
}

// local phase (+ the all-gather over the context's communicator when it has one) + the combine that needs no host decision
int job_enqueue(ncg_ctx* ctx, const JobRes& R, const ShardJob& J, bool collective, JobState* S) {
  const int G = collective && ctx->comm ? ctx->comm_size : 1;
  ShardJob job = J;
  if (collective) {
    job.part = ctx->comm ? ctx->comm_rank : 0;
    job.nparts = G;
  }
  int rc = job_local_phase(ctx, R, job, collective ? job.part : 0, G, S);
  if (rc) return rc;
  if (collective && ctx->comm) {  // a one-rank communicator runs the (in-place, trivial) all-gather too: same call path as G > 1
    const Rccl* r = rccl();
    if (!r) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: librccl.so not found");
    char* base = (char*)*R.comm_buf;
    // in place: rank r's slot already sits at offset r * stride of the receive buffer.  Jobs on the asynchronous lanes funnel
    // their collective through ONE stream per context (fork / join events around it): every collective of the communicator
    // is then enqueued on the same stream, in submit order, on every rank - no two all-gathers of one communicator are ever in
    // flight on different streams.
    hipStream_t cs = R.st;
    if (R.funnel) {
      if (!ctx->comm_stream) {
        NCG_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        NCG_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_fork, hipEventDisableTiming));
        NCG_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_join, hipEventDisableTiming));
      }
      cs = ctx->comm_stream;
      NCG_HIP(ctx, hipEventRecord(ctx->comm_fork, R.st));
      NCG_HIP(ctx, hipStreamWaitEvent(cs, ctx->comm_fork, 0));
    }
    NCG_NCCL(ctx, r, r->AllGather(base + S->stride * (size_t)ctx->comm_rank, base, S->stride, ncclUint8, (ncclComm_t)ctx->comm, cs));
    if (R.funnel) {
      NCG_HIP(ctx, hipEventRecord(ctx->comm_join, cs));
      NCG_HIP(ctx, hipStreamWaitEvent(R.st, ctx->comm_join, 0));
    }
  }
  return job_enqueue_combine(ctx, R, S, G);
}

// Before a synchronisation that is followed by job_finish_host: wait (polling, as hipStreamSynchronize would spin) until this
// rank's tail kernel has started, then wake the helper threads of the host finish - the rest of the tail, the exchange and the
// combine (0.1-0.2 ms) cover the 10-50 us a sleeping thread needs.  Without a flag (no local work on this rank) wake at once.
static void wait_tail_then_wake(const JobState& S, hipStream_t st) {
  if (S.tail_flag) {
    for (unsigned it = 0; *S.tail_flag != S.tail_gen; it++) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      if ((it & 127u) == 127u && hipStreamQuery(st) != hipErrorNotReady) break;
    }
  }
  ncg::msm_finish_prewake(S.curve);
}

int job_run_sync(ncg_ctx* ctx, const ShardJob& J, bool collective, void* out_affine, uint8_t* out_is_inf, hipStream_t st) {
  JobRes R = ctx_res(ctx, st);
  JobState S;
  int rc = job_enqueue(ctx, R, J, collective, &S);
  if (rc) {
    (void)hipStreamSynchronize(st);
    return rc;
  }
  wait_tail_then_wake(S, st);
  NCG_HIP(ctx, hipStreamSynchronize(st));
  return job_finish_host(ctx, R, S, out_affine, out_is_inf);
}

int check_common(ncg_ctx* ctx, int curve, const void* out_affine, const char* who) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (ncg_point_bytes(curve) == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: %s: unsupported curve %d", who, curve);
  if (!out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: %s: NULL output", who);
  return NCG_OK;
}
inline bool misaligned16(const void* p) { return ((uintptr_t)p & 15u) != 0; }

}  // namespace

struct ncg_multi {
  int n_dev = 0;
  std::vector<ncg_ctx*> ctx;
  std::vector<ncclComm_t> comms;
  std::string last_error;
};

#pragma GCC visibility push(default)
extern "C" {

int ncg_comm_unique_id(uint8_t* out_id128) {
  if (!out_id128) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: comm_unique_id: NULL buffer");
  const Rccl* r = rccl();
  if (!r) return set_err(nullptr, NCG_ERR_UNSUPPORTED, "noble-gpu: librccl.so not found (%s)", dlerror() ? dlerror() : "dlopen");
  ncclUniqueId id;
  NCG_NCCL(nullptr, r, r->GetUniqueId(&id));
  static_assert(sizeof id == NCG_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(out_id128, &id, sizeof id);
  return NCG_OK;
}

int ncg_comm_init(ncg_ctx* ctx, int nranks, int rank, const uint8_t* id128) {
  if (!ctx || !id128) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: comm_init: NULL argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: comm_init: bad rank %d / %d", rank, nranks);
  if (ctx->comm) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: comm_init: the context already has a communicator");
  const Rccl* r = rccl();
  if (!r) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: librccl.so not found");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclComm_t comm = nullptr;
  NCG_NCCL(ctx, r, r->CommInitRank(&comm, nranks, id, rank));
  ctx->comm = comm;
  ctx->comm_rank = rank;
  ctx->comm_size = nranks;
  return NCG_OK;
}

int ncg_comm_destroy(ncg_ctx* ctx) {
  if (!ctx || !ctx->comm) return NCG_OK;
  const Rccl* r = rccl();
  if (r) (void)r->CommDestroy((ncclComm_t)ctx->comm);
  ctx->comm = nullptr;
  ctx->comm_size = 1;
  ctx->comm_rank = 0;
  return NCG_OK;
}

int ncg_comm_size(ncg_ctx* ctx) { return ctx ? ctx->comm_size : 0; }
int ncg_comm_rank(ncg_ctx* ctx) { return ctx ? ctx->comm_rank : -1; }
// What the RCCL communicator itself reports (ncclCommCount / ncclCommUserRank), not the context's bookkeeping: the number of
// ranks the collectives of this context really span.  0 ranks = no communicator.  bench.py prints it as `rccl_ranks`.
int ncg_comm_count(ncg_ctx* ctx, int* out_ranks, int* out_rank) {
  if (!ctx || !out_ranks) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: comm_count: NULL argument");
  *out_ranks = 0;
  if (out_rank) *out_rank = -1;
  if (!ctx->comm) return NCG_OK;
  const Rccl* r = rccl();
  if (!r || !r->CommCount || !r->CommUserRank) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: comm_count: librccl.so has no ncclCommCount");
  int n = 0, me = -1;
  NCG_NCCL(ctx, r, r->CommCount((ncclComm_t)ctx->comm, &n));
  NCG_NCCL(ctx, r, r->CommUserRank((ncclComm_t)ctx->comm, &me));
  *out_ranks = n;
  if (out_rank) *out_rank = me;
  return NCG_OK;
}

int ncg_msm_sharded_dev(ncg_ctx* ctx, int curve, size_t n_local, size_t n_max, const void* points_affine_dev,
                        const void* scalars_dev, void* out_affine, uint8_t* out_is_inf, void* stream) {
  int rc = check_common(ctx, curve, out_affine, "msm_sharded");
  if (rc) return rc;
  if (n_max == 0) n_max = n_local;
  if (n_local > n_max || n_max > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_sharded: n_local > n_max");
  if (n_local && (!points_affine_dev || !scalars_dev)) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_sharded: NULL buffer");
  if (n_local && (misaligned16(points_affine_dev) || misaligned16(scalars_dev)))
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_sharded: device buffers must be 16-byte aligned");
  if (n_max == 0) return identity_out(curve, out_affine, out_is_inf);  // every shard empty (curve.ts:878)
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  ShardJob J;
  J.curve = curve;
  J.mode = ncg::SHARD_POINTS;
  J.n_local = n_local;
  J.n_plan = n_max;
  J.d_pts = points_affine_dev;
  J.d_sc = scalars_dev;
  return job_run_sync(ctx, J, true, out_affine, out_is_inf, st);
}

// ---- window-sharded mode: every rank holds ALL n points and scalars, rank r runs a contiguous range of the windows ----
static int windows_job(ncg_ctx* ctx, int curve, size_t n, const void* points_affine_dev, const ncg_points* resident, const void* scalars_dev,
                       ShardJob* J, const char* who) {
  if (resident) {
    if (resident->ctx != ctx) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: %s: handle does not belong to this context", who);
    curve = resident->curve;
    n = resident->n;
  }
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: %s: too many points", who);
  if (n && ((!points_affine_dev && !resident) || !scalars_dev)) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: %s: NULL buffer", who);
  if (n && ((points_affine_dev && misaligned16(points_affine_dev)) || misaligned16(scalars_dev)))
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: %s: device buffers must be 16-byte aligned", who);
  J->curve = curve;
  J->mode = ncg::SHARD_WINDOWS;
  J->n_local = J->n_plan = n;
  J->d_pts = resident ? nullptr : points_affine_dev;
  J->resident = resident;
  J->d_sc = scalars_dev;
  return NCG_OK;
}

int ncg_msm_sharded_windows_dev(ncg_ctx* ctx, int curve, size_t n, const void* points_affine_dev, const ncg_points* resident,
                                const void* scalars_dev, void* out_affine, uint8_t* out_is_inf, void* stream) {
  if (resident) curve = resident->curve;
  int rc = check_common(ctx, curve, out_affine, "msm_sharded_windows");
  if (rc) return rc;
  ShardJob J;
  rc = windows_job(ctx, curve, n, points_affine_dev, resident, scalars_dev, &J, "msm_sharded_windows");
  if (rc) return rc;
  if (J.n_plan == 0) return identity_out(J.curve, out_affine, out_is_inf);
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  return job_run_sync(ctx, J, true, out_affine, out_is_inf, stream ? (hipStream_t)stream : ctx->stream);
}

// ---- host-staged exchange: the same sharded MSM for transports other than RCCL (gloo, MPI, sockets) --------
// ncg_msm_shard_local_dev / ncg_msm_shard_windows_local_dev run this rank's phase and return its slot (header + grouped
// window sums, ncg_msm_shard_slot_bytes(curve) bytes) in host memory; the caller moves the slots of all ranks by whatever
// means it has and hands them to ncg_msm_shard_combine on any rank, which uploads them and runs the same header check,
// combine and finish that the RCCL entry points run after their all-gather.  A scalar outside the group order does NOT
// fail the local call: the verdict travels in the slot header and fails the combine on EVERY rank (a rank that raised
// before the exchange would leave the others waiting in it).
size_t ncg_msm_shard_slot_bytes(int curve) { return ncg_point_bytes(curve) ? ncg::msm_shard_slot_bytes(curve) : 0; }

static int local_to_host(ncg_ctx* ctx, const ShardJob& J, void* slot_out, hipStream_t st) {
  JobRes R = ctx_res(ctx, st);
  JobState S;
  int rc = job_local_phase(ctx, R, J, 0, 1, &S);
  if (rc) {
    (void)hipStreamSynchronize(st);
    return rc;
  }
  NCG_HIP_DRAIN(ctx, st, hipMemcpyAsync(*R.land, *R.comm_buf, S.stride, hipMemcpyDeviceToHost, st));
  NCG_HIP(ctx, hipStreamSynchronize(st));
  FinHeader h;
  memcpy(&h, *R.land, sizeof h);
  memset(slot_out, 0, S.stride);
  memcpy(slot_out, *R.land, sizeof h + (size_t)h.words * 4);
  return NCG_OK;
}

int ncg_msm_shard_local_dev(ncg_ctx* ctx, int curve, size_t n_local, size_t n_max, const void* points_affine_dev,
                            const void* scalars_dev, void* slot_out, void* stream) {
  int rc = check_common(ctx, curve, slot_out, "msm_shard_local");
  if (rc) return rc;
  if (n_max == 0) n_max = n_local;
  if (n_local > n_max || n_max > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_local: n_local > n_max");
  if (n_local && (!points_affine_dev || !scalars_dev)) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_local: NULL buffer");
  if (n_local && (misaligned16(points_affine_dev) || misaligned16(scalars_dev)))
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_local: device buffers must be 16-byte aligned");
  memset(slot_out, 0, ncg::msm_shard_slot_bytes(curve));
  if (n_max == 0) return NCG_OK;  // every shard empty: an all-zero slot (combine returns the identity for n_max = 0)
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  ShardJob J;
  J.curve = curve;
  J.mode = ncg::SHARD_POINTS;
  J.n_local = n_local;
  J.n_plan = n_max;
  J.d_pts = points_affine_dev;
  J.d_sc = scalars_dev;
  return local_to_host(ctx, J, slot_out, stream ? (hipStream_t)stream : ctx->stream);
}

int ncg_msm_shard_windows_local_dev(ncg_ctx* ctx, int curve, size_t n, int part, int nparts, const void* points_affine_dev,
                                    const ncg_points* resident, const void* scalars_dev, void* slot_out, void* stream) {
  if (resident) curve = resident->curve;
  int rc = check_common(ctx, curve, slot_out, "msm_shard_windows_local");
  if (rc) return rc;
  if (nparts < 1 || nparts > 4096 || part < 0 || part >= nparts) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_windows_local: bad part %d / %d", part, nparts);
  ShardJob J;
  rc = windows_job(ctx, curve, n, points_affine_dev, resident, scalars_dev, &J, "msm_shard_windows_local");
  if (rc) return rc;
  memset(slot_out, 0, ncg::msm_shard_slot_bytes(J.curve));
  if (J.n_plan == 0) return NCG_OK;
  J.part = part;
  J.nparts = nparts;
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  return local_to_host(ctx, J, slot_out, stream ? (hipStream_t)stream : ctx->stream);
}

int ncg_msm_shard_combine(ncg_ctx* ctx, int curve, size_t n_max, int nparts, const void* slots, void* out_affine, uint8_t* out_is_inf,
                          void* stream) {
  int rc = check_common(ctx, curve, out_affine, "msm_shard_combine");
  if (rc) return rc;
  if (!slots || nparts < 1 || nparts > 4096) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_combine: bad arguments");
  if (n_max > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_combine: too many points");
  if (n_max == 0) return identity_out(curve, out_affine, out_is_inf);  // every shard empty (curve.ts:878)
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const size_t stride = ncg::msm_shard_slot_bytes(curve);
  FinHeader h0;
  memcpy(&h0, slots, sizeof h0);
  JobRes R = ctx_res(ctx, st);
  JobState S;
  S.curve = curve;
  S.stride = stride;
  S.mode = h0.mode;
  if (h0.mode == ncg::SHARD_POINTS) {  // the plan this rank derives from n_max must be the one every slot was made with
    ncg::MsmPlan pl;
    if (ncg::msm_make_plan(curve, (int)n_max, 0, &pl) != 0) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: cannot plan windows");
    S.c = pl.c;
    S.nwin = pl.nwin;
  } else if (h0.mode == ncg::SHARD_WINDOWS || h0.mode == ncg::SHARD_WINDOWS_SHARED) {
    // window modes: the plan may be a resident set's own (endomorphism / precomputed): slot 0 names it, the check compares the rest
    if (h0.c < 2 || h0.c > 16 || h0.nwin < 1 || h0.nwin > 200) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_combine: malformed slot header");
    S.c = (int)h0.c;
    S.nwin = (int)h0.nwin;
  } else {
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_combine: malformed slot header (mode %u)", h0.mode);
  }
  rc = ensure_dev_buf(ctx, R.comm_buf, R.comm_buf_bytes, comm_buf_need(stride, ncg::msm_shard_max_fin_words(curve), nparts));
  if (rc) return rc;
  rc = ensure_land(ctx, R.land, R.land_words, (stride * (size_t)nparts + ncg::msm_shard_max_fin_words(curve) * 4) / 4 + 64);
  if (rc) return rc;
  // bound the payload sizes the device-side packing will read BEFORE trusting them (the full check runs in job_finish_host)
  for (int r = 0; r < nparts; r++) {
    FinHeader h;
    memcpy(&h, (const char*)slots + stride * (size_t)r, sizeof h);
    if ((size_t)h.words * 4 + sizeof h > stride) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_combine: malformed slot %d - all ranks must pass the same curve and n_max", r);
  }
  NCG_HIP(ctx, hipMemcpyAsync(*R.comm_buf, slots, stride * (size_t)nparts, hipMemcpyHostToDevice, st));
  rc = job_enqueue_combine(ctx, R, &S, nparts);
  if (rc) return rc;
  ncg::msm_finish_prewake(curve);   // the upload, the packing and the copy back are tens of microseconds: the helpers of the finish start now
  NCG_HIP(ctx, hipStreamSynchronize(st));
  return job_finish_host(ctx, R, S, out_affine, out_is_inf);
}

// The sharded pipelines on ONE GPU (self-check, shard-plan A/B, per-rank share measurements): the work is cut into
// `parts` as G ranks would cut it - slices of the points (ncg_msm_split_dev) or ranges of the windows
// (ncg_msm_split_windows_dev) - each part runs its local phase in turn, and the parts' slots go through the same combine
// and finish that G GPUs run after their all-gather.
static int split_run(ncg_ctx* ctx, ShardJob J, int parts, size_t n, int pb, void* out_affine, uint8_t* out_is_inf, hipStream_t st) {
  JobRes R = ctx_res(ctx, st);
  JobState S;
  const size_t per = (n + parts - 1) / parts;
  const char* pts0 = (const char*)J.d_pts;
  const char* sc0 = (const char*)J.d_sc;
  for (int g = 0; g < parts; g++) {
    if (J.mode == ncg::SHARD_POINTS) {
      const size_t lo = std::min(n, per * (size_t)g), cnt = std::min(n, lo + per) - lo;
      J.n_local = cnt;
      J.n_plan = per;
      J.d_pts = pts0 + lo * (size_t)pb;
      J.d_sc = sc0 + lo * 32;
    } else {
      J.part = g;
      J.nparts = parts;
    }
    int rc = job_local_phase(ctx, R, J, g, parts, &S);
    if (rc) {
      (void)hipStreamSynchronize(st);
      return rc;
    }
  }
  int rc = job_enqueue_combine(ctx, R, &S, parts);
  if (rc) return rc;
  wait_tail_then_wake(S, st);
  NCG_HIP(ctx, hipStreamSynchronize(st));
  if (J.mode == ncg::SHARD_POINTS) S.slice = per;
  return job_finish_host(ctx, R, S, out_affine, out_is_inf);
}

int ncg_msm_split_dev(ncg_ctx* ctx, int curve, size_t n, int parts, const void* points_affine_dev, const void* scalars_dev,
                      void* out_affine, uint8_t* out_is_inf, void* stream) {
  int rc = check_common(ctx, curve, out_affine, "msm_split");
  if (rc) return rc;
  const int pb = ncg_point_bytes(curve);
  if (parts < 1 || parts > 64) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_split: bad arguments");
  if (n == 0) return identity_out(curve, out_affine, out_is_inf);
  if (!points_affine_dev || !scalars_dev) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_split: NULL buffer");
  if (misaligned16(points_affine_dev) || misaligned16(scalars_dev))
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_split: device buffers must be 16-byte aligned");
  if ((n + parts - 1) / parts > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_split: too many points");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  ShardJob J;
  J.curve = curve;
  J.mode = ncg::SHARD_POINTS;
  J.d_pts = points_affine_dev;
  J.d_sc = scalars_dev;
  return split_run(ctx, J, parts, n, pb, out_affine, out_is_inf, stream ? (hipStream_t)stream : ctx->stream);
}

int ncg_msm_split_windows_dev(ncg_ctx* ctx, int curve, size_t n, int parts, const void* points_affine_dev, const ncg_points* resident,
                              const void* scalars_dev, void* out_affine, uint8_t* out_is_inf, void* stream) {
  if (resident) curve = resident->curve;
  int rc = check_common(ctx, curve, out_affine, "msm_split_windows");
  if (rc) return rc;
  if (parts < 1 || parts > 64) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_split_windows: bad arguments");
  ShardJob J;
  rc = windows_job(ctx, curve, n, points_affine_dev, resident, scalars_dev, &J, "msm_split_windows");
  if (rc) return rc;
  if (J.n_plan == 0) return identity_out(J.curve, out_affine, out_is_inf);
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  return split_run(ctx, J, parts, J.n_plan, ncg_point_bytes(J.curve), out_affine, out_is_inf, stream ? (hipStream_t)stream : ctx->stream);
}

// ---- several MSMs in flight ----------------------------------------------------------------------------------
// A lane owns a stream, a workspace, a gather buffer and a pinned landing area.  submit enqueues one MSM on the lane and
// returns; collect waits for it and runs the host finish.  With two or more lanes used in turn, the dependent tail of
// one MSM (narrow fold levels, per-window tail, D2H, host Horner - latency, not throughput) overlaps the sort and
// accumulate kernels of the next.  flags: NCG_MSM_ASYNC_WINDOWS = window-sharded over the context's communicator (a
// collective: every rank submits the same sequence of lanes).
static int lane_init(ncg_ctx* ctx, ncg_msm_lane& ln) {
  if (ln.stream) return NCG_OK;
  // The lanes must not share a hardware queue with one another: HIP spreads the streams of a process over
  // GPU_MAX_HW_QUEUES (4) queues PER PRIORITY CLASS, least-used first, so where lane 2 lands depends on every stream the
  // host application, torch and this library created before - in bench.py lanes 0 and 2 ended up behind one another and a
  // share of 0.50 ms read 0.59 (profiles/r06_lane_queues.txt).  The top-priority class is a pool of its own, which nothing
  // else here uses: the (at most 4) lanes get one queue each there, whatever the process did before.
  // NCG_LANE_QUEUES: 1 = that (default), 0 = plain streams, 2 = a full CU mask per lane stream (a queue of its own in the
  // normal class).
  const int qmode = ncg::knob("NCG_LANE_QUEUES", 1);
  hipError_t e = hipSuccess;
  if (qmode == 1) {
    int lo = 0, hi = 0;
    e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&ln.stream, hipStreamNonBlocking, hi);
  } else if (qmode == 2) {
    hipDeviceProp_t pr;
    e = hipGetDeviceProperties(&pr, ctx->device);
    if (e == hipSuccess) {
      std::vector<uint32_t> mask((size_t)(pr.multiProcessorCount + 31) / 32, 0xFFFFFFFFu);
      if (pr.multiProcessorCount % 32) mask.back() = (1u << (pr.multiProcessorCount % 32)) - 1u;
      e = hipExtStreamCreateWithCUMask(&ln.stream, (uint32_t)mask.size(), mask.data());
    }
  } else {
    e = hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking);
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ln.side.stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.side.fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.side.join, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.input_ready, hipEventDisableTiming);
  if (e != hipSuccess) return set_err(ctx, NCG_ERR_HIP, "noble-gpu: msm_async: cannot create the lane's stream: %s", hipGetErrorString(e));
  return NCG_OK;
}

int ncg_msm_async_lanes(void) { return NCG_MSM_LANES; }

int ncg_msm_async_submit(ncg_ctx* ctx, int lane, int curve, size_t n, const void* points_affine_dev, const ncg_points* resident,
                         const void* scalars_dev, int flags, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (lane < 0 || lane >= NCG_MSM_LANES) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_async: lane %d out of range (0..%d)", lane, NCG_MSM_LANES - 1);
  if (resident) curve = resident->curve;
  if (ncg_point_bytes(curve) == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: msm_async: unsupported curve %d", curve);
  ncg_msm_lane& ln = ctx->lanes[lane];
  if (ln.busy) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_async: lane %d has an MSM in flight - collect it first", lane);
  ShardJob J;
  int rc = windows_job(ctx, curve, n, points_affine_dev, resident, scalars_dev, &J, "msm_async");
  if (rc) return rc;
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  rc = lane_init(ctx, ln);
  if (rc) return rc;
  ln.curve = J.curve;
  ln.part_only = (flags & NCG_MSM_ASYNC_PART_FLAG) != 0;
  ln.state_identity = J.n_plan == 0;
  if (ln.state_identity) {
    ln.busy = true;
    return NCG_OK;
  }
  if (stream) {  // the inputs are produced on the caller's stream
    NCG_HIP(ctx, hipEventRecord(ln.input_ready, (hipStream_t)stream));
    NCG_HIP(ctx, hipStreamWaitEvent(ln.stream, ln.input_ready, 0));
  }
  JobRes R = lane_res(ln);
  JobState S;
  const bool collective = (flags & NCG_MSM_ASYNC_WINDOWS) != 0;
  if (ln.part_only) {  // one part of a window-sharded MSM; its slot comes back through ncg_msm_async_collect_slot
    J.part = (flags >> 8) & 0xFFF;
    J.nparts = (flags >> 20) & 0x7FF;
    if (collective || J.nparts < 1 || J.part >= J.nparts) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_async: bad part %d / %d", J.part, J.nparts);
    rc = job_local_phase(ctx, R, J, 0, 1, &S);
    if (rc == NCG_OK) {
      hipError_t e = hipMemcpyAsync(*R.land, *R.comm_buf, S.stride, hipMemcpyDeviceToHost, ln.stream);
      if (e != hipSuccess) rc = set_err(ctx, NCG_ERR_HIP, "noble-gpu: msm_async: %s", hipGetErrorString(e));
    }
  } else {
    rc = job_enqueue(ctx, R, J, collective, &S);  // not collective: one part holding every window
  }
  if (rc) {
    (void)hipStreamSynchronize(ln.stream);
    return rc;
  }
  NCG_HIP_DRAIN(ctx, ln.stream, hipEventRecord(ln.done, ln.stream));
  ln.mode = (int)S.mode;
  ln.nparts = S.nparts;
  ln.c = S.c;
  ln.nwin = S.nwin;
  ln.stride = S.stride;
  ln.busy = true;
  return NCG_OK;
}

int ncg_msm_async_collect(ncg_ctx* ctx, int lane, void* out_affine, uint8_t* out_is_inf) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (lane < 0 || lane >= NCG_MSM_LANES) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_async: lane %d out of range", lane);
  if (!out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_async: NULL output");
  ncg_msm_lane& ln = ctx->lanes[lane];
  if (!ln.busy) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_async: nothing was submitted on lane %d", lane);
  if (ln.part_only) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_async: lane %d holds one part of a sharded MSM - use ncg_msm_async_collect_slot", lane);
  ln.busy = false;
  if (ln.state_identity) return identity_out(ln.curve, out_affine, out_is_inf);
  NCG_HIP(ctx, hipEventSynchronize(ln.done));
  JobRes R = lane_res(ln);
  JobState S;
  S.curve = ln.curve;
  S.mode = (uint32_t)ln.mode;
  S.nparts = ln.nparts;
  S.c = ln.c;
  S.nwin = ln.nwin;
  S.stride = ln.stride;
  return job_finish_host(ctx, R, S, out_affine, out_is_inf);
}

int ncg_msm_async_collect_slot(ncg_ctx* ctx, int lane, void* slot_out) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (lane < 0 || lane >= NCG_MSM_LANES) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_async: lane %d out of range", lane);
  if (!slot_out) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_async: NULL output");
  ncg_msm_lane& ln = ctx->lanes[lane];
  if (!ln.busy || !ln.part_only) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_async: no part of a sharded MSM was submitted on lane %d", lane);
  ln.busy = false;
  ln.part_only = false;
  const size_t stride = ncg::msm_shard_slot_bytes(ln.curve);
  memset(slot_out, 0, stride);
  if (ln.state_identity) return NCG_OK;
  NCG_HIP(ctx, hipEventSynchronize(ln.done));
  FinHeader h;
  memcpy(&h, ln.land, sizeof h);
  if ((size_t)h.words * 4 + sizeof h > stride) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_async: malformed slot");
  memcpy(slot_out, ln.land, sizeof h + (size_t)h.words * 4);
  return NCG_OK;
}

/* ---- one process, several GPUs ---------------------------------------------------------------- */
int ncg_multi_init(const int* device_ids, int n_dev, ncg_multi** out) {
  if (!out || !device_ids || n_dev < 1 || n_dev > 64) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: multi_init: bad arguments");
  *out = nullptr;
  ncg_multi* m = new ncg_multi();
  m->n_dev = n_dev;
  m->ctx.assign(n_dev, nullptr);
  for (int i = 0; i < n_dev; i++) {
    int rc = ncg_init(device_ids[i], &m->ctx[i]);
    if (rc) {
      for (int j = 0; j < i; j++) ncg_destroy(m->ctx[j]);
      delete m;
      return rc;
    }
  }
  if (n_dev > 1) {
    const Rccl* r = rccl();
    ncclResult_t nr = ncclSuccess;
    m->comms.assign(n_dev, nullptr);
    if (!r || (nr = r->CommInitAll(m->comms.data(), n_dev, device_ids)) != ncclSuccess) {
      int rc = set_err(nullptr, r ? NCG_ERR_RCCL : NCG_ERR_UNSUPPORTED, "noble-gpu: multi_init: %s",
                       r ? r->GetErrorString(nr) : "librccl.so not found");
      for (int j = 0; j < n_dev; j++) ncg_destroy(m->ctx[j]);
      delete m;
      return rc;
    }
    for (int i = 0; i < n_dev; i++) {
      m->ctx[i]->comm = m->comms[i];
      m->ctx[i]->comm_rank = i;
      m->ctx[i]->comm_size = n_dev;
    }
  }
  *out = m;
  return NCG_OK;
}

void ncg_multi_destroy(ncg_multi* m) {
  if (!m) return;
  for (ncg_ctx* c : m->ctx) ncg_destroy(c);  // destroys each communicator with its context
  delete m;
}

int ncg_multi_devices(ncg_multi* m) { return m ? m->n_dev : 0; }
ncg_ctx* ncg_multi_ctx(ncg_multi* m, int i) { return (m && i >= 0 && i < m->n_dev) ? m->ctx[i] : nullptr; }
const char* ncg_multi_last_error(ncg_multi* m) { return m ? m->last_error.c_str() : ncg_last_error(nullptr); }

int ncg_msm_multi(ncg_multi* m, int curve, size_t n, const void* points_affine, const void* scalars, void* out_affine,
                  uint8_t* out_is_inf) {
  if (!m) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: multi is NULL");
  auto fail = [&](int rc) {
    m->last_error = ncg_last_error(nullptr);
    return rc;
  };
  const int pb = ncg_point_bytes(curve);
  if (pb == 0) return fail(set_err(nullptr, NCG_ERR_UNSUPPORTED, "noble-gpu: msm_multi: unsupported curve %d", curve));
  if (!out_affine) return fail(set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: msm_multi: NULL output"));
  if (n == 0) return identity_out(curve, out_affine, out_is_inf);
  if (!points_affine || !scalars) return fail(set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: msm_multi: NULL buffer"));
  const int G = m->n_dev;
  const size_t per = (n + G - 1) / G;
  if (per > 0x7fffffffu) return fail(set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: msm_multi: too many points"));
  const Rccl* r = G > 1 ? rccl() : nullptr;
  if (G > 1 && !r) return fail(set_err(nullptr, NCG_ERR_UNSUPPORTED, "noble-gpu: librccl.so not found"));
  std::vector<JobState> states(G);
  // 1. every device: upload its slice and run the device phase (all asynchronous, one host thread)
  for (int g = 0; g < G; g++) {
    ncg_ctx* ctx = m->ctx[g];
    const size_t lo = std::min(n, per * (size_t)g), cnt = std::min(n, lo + per) - lo;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(set_err(ctx, NCG_ERR_HIP, "noble-gpu: hipSetDevice failed"));
    const size_t pts_b = cnt * (size_t)pb, sc_b = cnt * 32;
    const size_t pts_al = (pts_b + 255) & ~(size_t)255;
    if (ctx->scratch_bytes < pts_al + sc_b + 512) {
      if (ctx->scratch) (void)hipFree(ctx->scratch);
      ctx->scratch = nullptr;
      ctx->scratch_bytes = 0;
      const size_t want = pts_al + sc_b + 4096;
      if (hipMalloc(&ctx->scratch, want) != hipSuccess) return fail(set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipMalloc(%zu) failed", want));
      ctx->scratch_bytes = want;
    }
    char* d_pts = (char*)ctx->scratch;
    char* d_sc = d_pts + pts_al;
    if (cnt) {
      if (hipMemcpyAsync(d_pts, (const char*)points_affine + lo * (size_t)pb, pts_b, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(d_sc, (const char*)scalars + lo * 32, sc_b, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(set_err(ctx, NCG_ERR_HIP, "noble-gpu: msm_multi: upload failed on device %d", ctx->device));
    }
    ShardJob J;
    J.curve = curve;
    J.mode = ncg::SHARD_POINTS;
    J.n_local = cnt;
    J.n_plan = per;
    J.d_pts = d_pts;
    J.d_sc = d_sc;
    int rc = job_local_phase(ctx, ctx_res(ctx, ctx->stream), J, g, G, &states[g]);
    if (rc) return fail(rc);
  }
  // 2. one grouped all-gather over the device set
  const size_t stride = states[0].stride;
  if (G > 1) {
    ncclResult_t nr = r->GroupStart();
    for (int g = 0; g < G && nr == ncclSuccess; g++) {
      ncg_ctx* ctx = m->ctx[g];
      char* base = (char*)ctx->comm_buf;
      nr = r->AllGather(base + stride * (size_t)g, base, stride, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream);
    }
    ncclResult_t ne = r->GroupEnd();
    if (nr != ncclSuccess || ne != ncclSuccess)
      return fail(set_err(nullptr, NCG_ERR_RCCL, "noble-gpu: msm_multi: RCCL all-gather failed (%s)",
                          r->GetErrorString(nr != ncclSuccess ? nr : ne)));
  }
  // 3. device 0 adds the partial arrays and finishes (the scalar verdicts of all devices travel in the slot headers);
  //    the others only drain their streams
  if (hipSetDevice(m->ctx[0]->device) != hipSuccess) return fail(set_err(m->ctx[0], NCG_ERR_HIP, "noble-gpu: hipSetDevice failed"));
  JobRes R0 = ctx_res(m->ctx[0], m->ctx[0]->stream);
  int rc = job_enqueue_combine(m->ctx[0], R0, &states[0], G);
  for (int g = 0; g < G; g++) {
    (void)hipSetDevice(m->ctx[g]->device);
    (void)hipStreamSynchronize(m->ctx[g]->stream);
  }
  (void)hipSetDevice(m->ctx[0]->device);
  if (rc) return fail(rc);
  states[0].slice = per;
  rc = job_finish_host(m->ctx[0], R0, states[0], out_affine, out_is_inf);
  if (rc) return fail(rc);
  return NCG_OK;
}

}  // extern "C"
#pragma GCC visibility pop
