// C ABI (include/ncg.h): context, workspace and dispatch.  No torch types cross here.
#include <hip/hip_runtime.h>
#include "knobs.hpp"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ncg.h"
#include "host_api.hpp"
#include "consts_gen.hpp"
#include "msm.hpp"

namespace {
std::mutex g_err_mu;
std::string g_last_error;
}  // namespace

#include "ctx.hpp"

int ncg_set_err(ncg_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  {
    std::lock_guard<std::mutex> g(g_err_mu);
    g_last_error = buf;
  }
  if (ctx) ctx->last_error = buf;
  return code;
}

// Host <-> device copies of the host-pointer entry points.  Pageable user buffers move at a few
// GB/s through the runtime's staging path; registering (pinning) a large buffer for the duration of
// the call lets the DMA engines read/write it directly at PCIe speed.  Everything registered here is
// released when the object goes out of scope, after the stream has been drained.
struct PinSet {
  ncg_ctx* ctx;
  struct Range { uintptr_t lo, hi; };
  Range regs[24];
  int n = 0;
  explicit PinSet(ncg_ctx* c) : ctx(c) {}
  // Page-locks [p, p + bytes) for the duration of the call.  A copy must lie inside ONE registration, and two registrations
  // must not share a page: callers that pin an array piecewise cut it at page boundaries (PagedParts below).
  // Buffers the caller registered itself (ncg_host_register) fail here and are left alone.
  // (The EXACT range is registered, not its pages: the runtime treats a pointer as pinned by the registered range, and a range
  // widened to its pages would claim the head of whatever the caller allocated next to the buffer - a copy into that neighbour
  // would then start inside a registration and run out of it: hipErrorInvalidValue.)
  void pin(const void* p, size_t bytes) {
    if (bytes < ((size_t)1 << 20) || n >= 24) return;
    {  // already pinned (ncg_host_register, hipHostMalloc): a second registration of a PIECE of it would succeed and cost the
       // page locking again - ask first
      // (both ends: a buffer registered only as a PREFIX of this piece is not "already pinned" - it is left to the plain copy,
      // which reports the partial registration instead of a DMA running past it)
      hipPointerAttribute_t at;
      if (hipPointerGetAttributes(&at, p) == hipSuccess) {
        if (at.type == hipMemoryTypeHost) return;
      } else {
        (void)hipGetLastError();  // plain pageable memory is "invalid value" to this query on some runtimes
      }
      if (hipPointerGetAttributes(&at, (const char*)p + bytes - 1) == hipSuccess) {
        if (at.type == hipMemoryTypeHost) return;   // the tail belongs to someone's registration: registering the range would overlap it
      } else {
        (void)hipGetLastError();
      }
    }
    if (hipHostRegister((void*)p, bytes, hipHostRegisterDefault) == hipSuccess) regs[n++] = Range{(uintptr_t)p, (uintptr_t)p + bytes};
    else (void)hipGetLastError();  // not registrable: plain copy
  }
  hipError_t h2d(void* dst, const void* src, size_t bytes) {
    pin(src, bytes);
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
  }
  hipError_t d2h(void* dst, const void* src, size_t bytes) {
    pin(dst, bytes);
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
  }
  ~PinSet() {
    if (n) {  // nothing may still be reading or writing the buffers when they are unpinned - on ANY path out of the caller
      (void)hipStreamSynchronize(ctx->stream);
      if (ctx->copy_in) (void)hipStreamSynchronize(ctx->copy_in);
      if (ctx->copy_out) (void)hipStreamSynchronize(ctx->copy_out);
    }
    for (int i = 0; i < n; i++) (void)hipHostUnregister((void*)regs[i].lo);
  }
};

// One host array crossing the bus piecewise: consecutive byte ranges whose INNER boundaries are page boundaries of the host
// address space, so that each piece can be page-locked on its own just before its copy is enqueued (the host locks piece p + 1
// while piece p is on the bus; locking 128 MB up front costs ~3 ms before the first byte moves) and no two registrations share a
// page.  Uploads round a piece's end UP (a few bytes of the next piece travel early); downloads round it DOWN (the last partial
// page of a piece is fetched with the next one, when all of it has been computed).
struct PagedParts {
  const char* base;
  size_t total, done = 0;
  PagedParts(const void* b, size_t t) : base((const char*)b), total(t) {}
  // piece that makes bytes [0, need_end) available (upload) / that may fetch everything below need_end (download)
  bool next(size_t need_end, bool last, bool round_up, size_t* lo, size_t* hi) {
    const uintptr_t a = (uintptr_t)base + need_end;
    size_t e = last ? total : (size_t)(((round_up ? a + 4095 : a) & ~(uintptr_t)4095) - (uintptr_t)base);
    if (!last && (a & ~(uintptr_t)4095) < (uintptr_t)base) e = 0;   // (download, first page before the array)
    e = std::min(e, total);
    if (e <= done) return false;
    *lo = done;
    *hi = e;
    done = e;
    return true;
  }
};

static int ensure_scratch(ncg_ctx* ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return NCG_OK;
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  size_t want = bytes + (bytes >> 2) + 4096;
  hipError_t e = hipMalloc(&ctx->scratch, want);
  if (e != hipSuccess) return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  ctx->scratch_bytes = want;
  return NCG_OK;
}

// window plan for an n-point MSM (c_override > 0 fixes the window width) and a workspace big enough for it
static int msm_ensure_ws(ncg_ctx* ctx, int curve, ncg::MsmPlan& pl);
int ncg_msm_plan_ws(ncg_ctx* ctx, int curve, size_t n, int c_override, ncg::MsmPlan* pl) {
  if (ncg::msm_make_plan(curve, (int)n, c_override, pl) != 0)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: cannot plan windows");
  return msm_ensure_ws(ctx, curve, *pl);
}
// the context's tuning overrides and trace slot go into the plan (every MSM entry point passes through here or
// through ncg_msm_ensure_buf), then the workspace grows if the plan needs more
static void msm_apply_ctx(ncg_ctx* ctx, ncg::MsmPlan& pl) {
  pl.seg_override = ctx->msm_seg_override;
  pl.run_serial_override = ctx->msm_run_serial_override;
  pl.trace = &ctx->msm_trace;
}
int ncg_msm_ensure_buf(ncg_ctx* ctx, int curve, ncg::MsmPlan& pl, void** ws, size_t* ws_bytes) {
  msm_apply_ctx(ctx, pl);
  size_t need = ncg::msm_workspace_bytes(curve, pl);
  if (*ws_bytes < need) {
    // ncg_msm_last_plan reads the long-run counter of the last MSM out of ITS workspace: forget the pointer when that
    // workspace is the one being replaced (the trace then reports 0 runs instead of reading freed memory)
    const char* lr = (const char*)ctx->msm_trace.d_long_runs;
    if (*ws && lr && lr >= (const char*)*ws && lr < (const char*)*ws + *ws_bytes) ctx->msm_trace.d_long_runs = nullptr;
    if (*ws) (void)hipFree(*ws);
    *ws = nullptr;
    *ws_bytes = 0;
    hipError_t e = hipMalloc(ws, need);
    if (e != hipSuccess)
      return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: msm workspace hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
    *ws_bytes = need;
  }
  return NCG_OK;
}
static int msm_ensure_ws(ncg_ctx* ctx, int curve, ncg::MsmPlan& pl) { return ncg_msm_ensure_buf(ctx, curve, pl, &ctx->msm_ws, &ctx->msm_ws_bytes); }
int ncg_msm_plan_ws_windows(ncg_ctx* ctx, int curve, size_t n, int w0, int cnt, ncg::MsmPlan* pl, void** ws, size_t* ws_bytes) {
  if (ncg::msm_make_plan(curve, (int)n, 0, pl) != 0)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: cannot plan windows");
  if (w0 < 0 || cnt < 0 || w0 + cnt > pl->nwin) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: window range outside the plan");
  ncg::msm_plan_take_windows(*pl, w0, cnt);
  return ncg_msm_ensure_buf(ctx, curve, *pl, ws ? ws : &ctx->msm_ws, ws_bytes ? ws_bytes : &ctx->msm_ws_bytes);
}
// device buffers are read with 16-byte accesses (ncg.h "Conventions"): refuse a misaligned pointer instead of faulting
static inline bool misaligned16(const void* p) { return ((uintptr_t)p & 15u) != 0; }

// only the C ABI of include/ncg.h is exported (the objects are built with -fvisibility=hidden)
#pragma GCC visibility push(default)
extern "C" {

const char* ncg_version(void) { return "noble-curves-amd 0.1 (gfx950)"; }

int ncg_point_bytes(int curve) {
  switch (curve) {
    case NCG_SECP256K1: return 64;
    case NCG_ED25519: return 64;
    case NCG_BLS12_381_G1: return 96;
    case NCG_BLS12_381_G2: return 192;
    default: return 0;
  }
}
int ncg_field_bytes(int curve) { return ncg_point_bytes(curve) / 2; }

int ncg_init(int device_id, ncg_ctx** out_ctx) {
  if (!out_ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: out_ctx is NULL");
  *out_ctx = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return set_err(nullptr, NCG_ERR_NO_DEVICE, "noble-gpu: no HIP device visible (%s)", hipGetErrorString(e));
  if (device_id < 0 || device_id >= count)
    return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: device %d out of range (have %d)", device_id, count);
  ncg_ctx* ctx = new ncg_ctx();
  ctx->device = device_id;
  e = hipSetDevice(device_id);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->msm_side.stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->msm_side.fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->msm_side.join, hipEventDisableTiming);
  if (e != hipSuccess) {
    int rc = set_err(nullptr, NCG_ERR_HIP, "noble-gpu: cannot create stream on device %d: %s", device_id, hipGetErrorString(e));
    ncg_destroy(ctx);  // releases whichever streams / events were created before the failure
    return rc;
  }
  *out_ctx = ctx;
  return NCG_OK;
}

void ncg_destroy(ncg_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->msm_ws) (void)hipFree(ctx->msm_ws);
  if (ctx->mul_ws) (void)hipFree(ctx->mul_ws);
  if (ctx->ed_btab) (void)hipFree(ctx->ed_btab);
  if (ctx->ed_ks) (void)hipFree(ctx->ed_ks);
  if (ctx->ecdsa_ws) (void)hipFree(ctx->ecdsa_ws);
  for (int i = 0; i < 4; i++)
    if (ctx->base_tab[i]) (void)hipFree(ctx->base_tab[i]);
  if (ctx->ub_in) (void)hipFree(ctx->ub_in);
  if (ctx->ub_out) (void)hipFree(ctx->ub_out);
  for (int i = 0; i <= NCG_NTT_MAX_LOG2N; i++)
    if (ctx->ntt_tab[i]) (void)hipFree(ctx->ntt_tab[i]);
  if (ctx->ntt_ws) (void)hipFree(ctx->ntt_ws);
  (void)ncg_comm_destroy(ctx);
  if (ctx->comm_buf) (void)hipFree(ctx->comm_buf);
  if (ctx->sync_land) (void)hipHostFree(ctx->sync_land);
  if (ctx->comm_fork) (void)hipEventDestroy(ctx->comm_fork);
  if (ctx->comm_join) (void)hipEventDestroy(ctx->comm_join);
  if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
  for (int i = 0; i < ncg_ctx::COPY_CHUNKS; i++) {
    if (ctx->ev_in[i]) (void)hipEventDestroy(ctx->ev_in[i]);
    if (ctx->ev_k[i]) (void)hipEventDestroy(ctx->ev_k[i]);
    if (ctx->ev_sc[i]) (void)hipEventDestroy(ctx->ev_sc[i]);
  }
  if (ctx->ev_ready) (void)hipEventDestroy(ctx->ev_ready);
  if (ctx->copy_in) (void)hipStreamDestroy(ctx->copy_in);
  if (ctx->copy_out) (void)hipStreamDestroy(ctx->copy_out);
  for (ncg_msm_lane& ln : ctx->lanes) {
    if (ln.stream) (void)hipStreamSynchronize(ln.stream);
    if (ln.ws) (void)hipFree(ln.ws);
    if (ln.comm_buf) (void)hipFree(ln.comm_buf);
    if (ln.land) (void)hipHostFree(ln.land);
    if (ln.done) (void)hipEventDestroy(ln.done);
    if (ln.input_ready) (void)hipEventDestroy(ln.input_ready);
    if (ln.side.fork) (void)hipEventDestroy(ln.side.fork);
    if (ln.side.join) (void)hipEventDestroy(ln.side.join);
    if (ln.side.stream) (void)hipStreamDestroy(ln.side.stream);
    if (ln.stream) (void)hipStreamDestroy(ln.stream);
  }
  if (ctx->msm_side.fork) (void)hipEventDestroy(ctx->msm_side.fork);
  if (ctx->msm_side.join) (void)hipEventDestroy(ctx->msm_side.join);
  if (ctx->msm_side.stream) (void)hipStreamDestroy(ctx->msm_side.stream);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* ncg_last_error(ncg_ctx* ctx) {
  if (ctx) return ctx->last_error.c_str();
  std::lock_guard<std::mutex> g(g_err_mu);
  static thread_local std::string copy;
  copy = g_last_error;
  return copy.c_str();
}

int ncg_sync(ncg_ctx* ctx) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

static int ensure_mul_ws(ncg_ctx* ctx, int curve, size_t n, hipStream_t st) {
  size_t need = ncg::mul_var_tmp_bytes(curve, (int)n);
  if (ctx->mul_ws_bytes >= need) return NCG_OK;
  NCG_HIP(ctx, hipStreamSynchronize(st));
  if (ctx->mul_ws) (void)hipFree(ctx->mul_ws);
  ctx->mul_ws = nullptr;
  ctx->mul_ws_bytes = 0;
  hipError_t e = hipMalloc(&ctx->mul_ws, need);
  if (e != hipSuccess) return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
  ctx->mul_ws_bytes = need;
  // touch the fresh allocation once here, so the first kernel that uses it is not the one paying for
  // the page mappings of a GB-sized buffer
  NCG_HIP(ctx, hipMemsetAsync(ctx->mul_ws, 0, need, st));
  return NCG_OK;
}

int ncg_mul_var_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* points_affine_dev, const void* scalars_dev,
                          void* out_affine_dev, uint8_t* out_is_inf_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (curve < NCG_SECP256K1 || curve > NCG_BLS12_381_G2)
    return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: mul_var_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!points_affine_dev || !scalars_dev || !out_affine_dev || !out_is_inf_dev)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: mul_var_batch: NULL buffer");
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  {
    int rc = ensure_mul_ws(ctx, curve, n, st);
    if (rc) return rc;
  }
  NCG_HIP(ctx, ncg::mul_var_batch(curve, (const uint32_t*)points_affine_dev, (const uint32_t*)scalars_dev,
                                  (uint32_t*)out_affine_dev, out_is_inf_dev, (int)n, (uint32_t*)ctx->mul_ws, st));
  return NCG_OK;
}

// Long-lived host buffers (the Node addon's Buffers, a prover's witness arrays) can be pinned ONCE: the host-pointer entry
// points then DMA straight from / to them.  Without it every call registers the large buffers it is handed and releases
// them again (PinSet): correct, but the page locking is paid per call.
int ncg_host_register(void* p, size_t bytes) {
  if (!p || !bytes) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: host_register: NULL buffer");
  hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return set_err(nullptr, NCG_ERR_HIP, "noble-gpu: host_register: %s", hipGetErrorString(e));
  }
  return NCG_OK;
}
int ncg_host_unregister(void* p) {
  if (!p) return NCG_OK;
  hipError_t e = hipHostUnregister(p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return set_err(nullptr, NCG_ERR_HIP, "noble-gpu: host_unregister: %s", hipGetErrorString(e));
  }
  return NCG_OK;
}

// copy streams + chunk events of the host-pointer entry points (made on first use)
static int ensure_copy_streams(ncg_ctx* ctx) {
  if (ctx->copy_in) return NCG_OK;
  hipError_t e = hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking);
  for (int i = 0; i < ncg_ctx::COPY_CHUNKS && e == hipSuccess; i++) {
    e = hipEventCreateWithFlags(&ctx->ev_in[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_k[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_sc[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming);
  if (e != hipSuccess) return set_err(ctx, NCG_ERR_HIP, "noble-gpu: cannot create copy streams: %s", hipGetErrorString(e));
  return NCG_OK;
}
// Waits for every stream a host-pointer call may have used; returns the FIRST failure (an asynchronous copy or kernel error
// surfaces here, not at enqueue time), after having waited on all of them.
static hipError_t drain_copy_streams(ncg_ctx* ctx) {
  hipError_t first = hipSuccess, e;
  if (ctx->copy_in && (e = hipStreamSynchronize(ctx->copy_in)) != hipSuccess && first == hipSuccess) first = e;
  if (ctx->copy_out && (e = hipStreamSynchronize(ctx->copy_out)) != hipSuccess && first == hipSuccess) first = e;
  if (ctx->msm_side.stream && (e = hipStreamSynchronize(ctx->msm_side.stream)) != hipSuccess && first == hipSuccess) first = e;
  if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess && first == hipSuccess) first = e;
  return first;
}

int ncg_mul_var_batch(ncg_ctx* ctx, int curve, size_t n, const void* points_affine, const void* scalars,
                      void* out_affine, uint8_t* out_is_inf) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  int pb = ncg_point_bytes(curve);
  if (pb == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: mul_var_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (!points_affine || !scalars || !out_affine)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: mul_var_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  size_t pts_b = n * pb, sc_b = n * 32, inf_b = (n + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, 2 * pts_b + sc_b + inf_b + 1024);
  if (rc) return rc;
  char* base = (char*)ctx->scratch;
  char* d_pts = base;
  char* d_out = d_pts + pts_b;
  char* d_sc = d_out + pts_b;
  char* d_inf = d_sc + sc_b;
  if (n >= ((size_t)1 << 17)) {
    // Large batches in chunks: chunk i + 1 crosses PCIe while the kernels of chunk i run and the results of chunk i - 1
    // go back (three streams, one event per chunk and direction).  The kernels are the batch's whole cost but for the
    // first upload and the last download: 2^20 secp256k1 pairs 13.4 -> ~9.6 ms end to end.
    rc = ensure_copy_streams(ctx);
    if (rc) return rc;
    // (page locking per chunk, just ahead of its copies - see ncg_msm)
    // Chunk sizes 1 : 3 : 3 : 1 (in eighths of the batch) for large batches: a launch of the ladder pays ~0.3 ms of ramp whatever
    // its size (tools/secp_rounds.py: 9.8 ns per item for one round of 196 608 items, 8.6 for two, 8.3 for the whole batch), so the
    // middle chunks are big, and only the first upload and the last download are exposed, so the outer ones are small.
    // (eight equal chunks: 10.7 ms for 2^20 secp256k1 pairs from pinned memory.)
    static const int k_eighths_big[4] = {1, 3, 3, 1}, k_eighths_even[4] = {2, 2, 2, 2};
    const int* eighths = (n >= ((size_t)1 << 19) && ncg::knob("NCG_MULVAR_HOST_EVEN", 0) == 0) ? k_eighths_big : k_eighths_even;  // (knob: A/B builds only)
    const int chunks = 4;
    const size_t unit = (((n + 7) / 8) + 255) & ~(size_t)255;
    hipError_t e = hipSuccess;
    size_t lo = 0;
    PagedParts in_pts(points_affine, pts_b), in_sc(scalars, sc_b), out_pts(out_affine, pts_b), out_inf(out_is_inf, n);
    for (int c = 0; c < chunks && e == hipSuccess && rc == NCG_OK; c++) {
      const size_t cnt = std::min(n - lo, unit * (size_t)eighths[c]);
      if (cnt == 0) break;
      const bool last = lo + cnt >= n;
      size_t a, b;
      if (in_pts.next((lo + cnt) * pb, last, true, &a, &b)) {
        pins.pin((const char*)points_affine + a, b - a);
        e = hipMemcpyAsync(d_pts + a, (const char*)points_affine + a, b - a, hipMemcpyHostToDevice, ctx->copy_in);
      }
      if (e == hipSuccess && in_sc.next((lo + cnt) * 32, last, true, &a, &b)) {
        pins.pin((const char*)scalars + a, b - a);
        e = hipMemcpyAsync(d_sc + a, (const char*)scalars + a, b - a, hipMemcpyHostToDevice, ctx->copy_in);
      }
      if (e == hipSuccess) e = hipEventRecord(ctx->ev_in[c], ctx->copy_in);
      if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->ev_in[c], 0);
      if (e != hipSuccess) break;
      rc = ncg_mul_var_batch_dev(ctx, curve, cnt, d_pts + lo * pb, d_sc + lo * 32, d_out + lo * pb, (uint8_t*)d_inf + lo, ctx->stream);
      if (rc) break;
      e = hipEventRecord(ctx->ev_k[c], ctx->stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_out, ctx->ev_k[c], 0);
      if (e == hipSuccess && out_pts.next((lo + cnt) * pb, last, false, &a, &b)) {
        pins.pin((char*)out_affine + a, b - a);   // (locked while the chunk's kernels run)
        e = hipMemcpyAsync((char*)out_affine + a, d_out + a, b - a, hipMemcpyDeviceToHost, ctx->copy_out);
      }
      if (e == hipSuccess && out_is_inf && out_inf.next(lo + cnt, last, false, &a, &b)) {
        pins.pin(out_is_inf + a, b - a);
        e = hipMemcpyAsync(out_is_inf + a, d_inf + a, b - a, hipMemcpyDeviceToHost, ctx->copy_out);
      }
      lo += cnt;
    }
    const hipError_t ed = drain_copy_streams(ctx);
    if (rc) return rc;
    if (e == hipSuccess) e = ed;
    if (e != hipSuccess) return set_err(ctx, NCG_ERR_HIP, "noble-gpu: mul_var_batch: %s", hipGetErrorString(e));
    return NCG_OK;
  }
  NCG_HIP(ctx, pins.h2d(d_pts, points_affine, pts_b));
  NCG_HIP(ctx, pins.h2d(d_sc, scalars, sc_b));
  rc = ncg_mul_var_batch_dev(ctx, curve, n, d_pts, d_sc, d_out, (uint8_t*)d_inf, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_affine, d_out, pts_b));
  if (out_is_inf) NCG_HIP(ctx, pins.d2h(out_is_inf, d_inf, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

int ncg_add_pairs_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* a_dev, const void* b_dev, int subtract,
                            void* out_affine_dev, uint8_t* out_is_inf_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (curve < NCG_SECP256K1 || curve > NCG_BLS12_381_G2)
    return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: add_pairs_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!a_dev || !b_dev || !out_affine_dev || !out_is_inf_dev)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: add_pairs_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  int rc = ensure_mul_ws(ctx, curve, n, st);
  if (rc) return rc;
  NCG_HIP(ctx, ncg::pair_add_batch(curve, (const uint32_t*)a_dev, (const uint32_t*)b_dev, subtract, (uint32_t*)out_affine_dev,
                                   out_is_inf_dev, (int)n, (uint32_t*)ctx->mul_ws, st));
  return NCG_OK;
}

int ncg_add_pairs_batch(ncg_ctx* ctx, int curve, size_t n, const void* a, const void* b, int subtract, void* out_affine,
                        uint8_t* out_is_inf) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  int pb = ncg_point_bytes(curve);
  if (pb == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: add_pairs_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (!a || !b || !out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: add_pairs_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  size_t pts_b = n * (size_t)pb, inf_b = (n + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, 3 * pts_b + inf_b + 1024);
  if (rc) return rc;
  char* d_a = (char*)ctx->scratch;
  char* d_b = d_a + pts_b;
  char* d_out = d_b + pts_b;
  char* d_inf = d_out + pts_b;
  NCG_HIP(ctx, pins.h2d(d_a, a, pts_b));
  NCG_HIP(ctx, pins.h2d(d_b, b, pts_b));
  rc = ncg_add_pairs_batch_dev(ctx, curve, n, d_a, d_b, subtract, d_out, (uint8_t*)d_inf, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_affine, d_out, pts_b));
  if (out_is_inf) NCG_HIP(ctx, pins.d2h(out_is_inf, d_inf, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

int ncg_mul_base_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* scalars_dev, void* out_affine_dev,
                           uint8_t* out_is_inf_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (curve < NCG_SECP256K1 || curve > NCG_BLS12_381_G2)
    return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: mul_base_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!scalars_dev || !out_affine_dev || !out_is_inf_dev)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: mul_base_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  int rc = ensure_mul_ws(ctx, curve, n > 8192 ? n : 8192, st);
  if (rc) return rc;
  if (curve == NCG_ED25519) {  // table computed on the host (33 x 128 affine Niels points), cached per context
    if (!ctx->base_tab[curve]) {
      std::vector<uint32_t> host(ncg::ed25519_fixed_table_words());
      ncg::ed25519_build_fixed_table(host.data());
      uint32_t* tab = nullptr;
      NCG_HIP(ctx, hipMalloc((void**)&tab, host.size() * 4));
      hipError_t e = hipMemcpy(tab, host.data(), host.size() * 4, hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        (void)hipFree(tab);
        return set_err(ctx, NCG_ERR_HIP, "noble-gpu: uploading the ed25519 fixed-base table failed: %s", hipGetErrorString(e));
      }
      ctx->base_tab[curve] = tab;
    }
    NCG_HIP(ctx, ncg::ed25519_mul_base_batch(ctx->base_tab[curve], (const uint32_t*)scalars_dev, (uint32_t*)out_affine_dev,
                                             out_is_inf_dev, (int)n, (uint32_t*)ctx->mul_ws, st));
    return NCG_OK;
  }
  if (!ctx->base_tab[curve]) {  // built once per context with the variable-base kernel
    const uint32_t* base = curve == NCG_SECP256K1 ? ncg::BasePoints::SECP
                           : curve == NCG_BLS12_381_G1 ? ncg::BasePoints::G1 : ncg::BasePoints::G2;
    uint32_t* tab = nullptr;
    NCG_HIP(ctx, hipMalloc((void**)&tab, ncg::mul_base_table_bytes(curve)));
    hipError_t e = ncg::mul_base_build_table(curve, base, tab, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // cache the table only once the build is known to have completed
    if (e != hipSuccess) {
      (void)hipFree(tab);
      return set_err(ctx, NCG_ERR_HIP, "noble-gpu: building the fixed-base table failed: %s", hipGetErrorString(e));
    }
    ctx->base_tab[curve] = tab;
  }
  NCG_HIP(ctx, ncg::mul_base_batch(curve, ctx->base_tab[curve], (const uint32_t*)scalars_dev, (uint32_t*)out_affine_dev,
                                   out_is_inf_dev, (int)n, (uint32_t*)ctx->mul_ws, st));
  return NCG_OK;
}

int ncg_mul_base_batch(ncg_ctx* ctx, int curve, size_t n, const void* scalars, void* out_affine, uint8_t* out_is_inf) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  int pb = ncg_point_bytes(curve);
  if (pb == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: mul_base_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (!scalars || !out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: mul_base_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  size_t pts_b = n * pb, sc_b = n * 32, inf_b = (n + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, pts_b + sc_b + inf_b + 1024);
  if (rc) return rc;
  char* d_out = (char*)ctx->scratch;
  char* d_sc = d_out + pts_b;
  char* d_inf = d_sc + sc_b;
  NCG_HIP(ctx, pins.h2d(d_sc, scalars, sc_b));
  rc = ncg_mul_base_batch_dev(ctx, curve, n, d_sc, d_out, (uint8_t*)d_inf, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_affine, d_out, pts_b));
  if (out_is_inf) NCG_HIP(ctx, pins.d2h(out_is_inf, d_inf, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

// window plan the MSM entry points use for n points: out = {c, nwin, buckets per window, grouped sums per window}
int ncg_msm_plan_info(int curve, size_t n, int* out4) {
  if (!out4 || ncg_point_bytes(curve) == 0 || n == 0 || n > 0x7fffffffu) return NCG_ERR_INVALID_ARG;
  ncg::MsmPlan pl;
  if (ncg::msm_make_plan(curve, (int)n, 0, &pl) != 0) return NCG_ERR_INVALID_ARG;
  out4[0] = pl.c;
  out4[1] = pl.nwin;
  out4[2] = pl.nb;
  out4[3] = (int)(ncg::msm_fin_words(curve, pl) / ncg::msm_acc_words(curve) / (size_t)pl.nwin);
  return NCG_OK;
}

// Tuning overrides of the MSM on this context (diagnostics / tests: the lane segment decides how the accumulate kernel
// cuts buckets, run_serial which cut buckets go to the long-run work list).  seg <= 0 and run_serial < 0 restore the defaults.
int ncg_msm_set_tuning(ncg_ctx* ctx, int seg, int run_serial) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (seg > (1 << 24) || run_serial > (1 << 24)) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_set_tuning: value out of range");
  ctx->msm_seg_override = seg > 0 ? seg : 0;
  ctx->msm_run_serial_override = run_serial >= 0 ? run_serial : -1;
  return NCG_OK;
}
// What the last MSM launch on this context ran with: out8 = {c, local windows, buckets per window, first window, windows of
// the whole plan, entries per accumulate lane (seg), run_serial, runs that went to the long-run work list}.  Synchronises.
int ncg_msm_last_plan(ncg_ctx* ctx, int* out8) {
  if (!ctx || !out8) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_last_plan: NULL argument");
  const ncg::MsmTrace& tr = ctx->msm_trace;
  if (tr.c == 0) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_last_plan: no MSM has run on this context");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  NCG_HIP(ctx, hipDeviceSynchronize());
  uint32_t runs = 0;
  if (tr.d_long_runs) NCG_HIP(ctx, hipMemcpy(&runs, tr.d_long_runs, 4, hipMemcpyDeviceToHost));
  out8[0] = tr.c; out8[1] = tr.nwin; out8[2] = tr.nb; out8[3] = tr.w0; out8[4] = tr.nwin_total;
  out8[5] = tr.seg; out8[6] = tr.run_serial; out8[7] = (int)runs;
  return NCG_OK;
}

int ncg_msm_dev(ncg_ctx* ctx, int curve, size_t n, const void* points_affine_dev, const void* scalars_dev,
                void* out_affine, uint8_t* out_is_inf, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  int pb = ncg_point_bytes(curve);
  if (pb == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: msm: unsupported curve %d", curve);
  if (!out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: NULL output");
  uint8_t inf_local = 0;
  if (n == 0) {  // empty MSM is the identity (reference curve.ts:878)
    memset(out_affine, 0, pb);
    if (curve == NCG_ED25519) ((uint8_t*)out_affine)[32] = 1;  // Edwards identity is (0, 1)
    if (out_is_inf) *out_is_inf = 1;
    return NCG_OK;
  }
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: too many points");
  if (!points_affine_dev || !scalars_dev) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: NULL buffer");
  if (misaligned16(points_affine_dev) || misaligned16(scalars_dev))
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: device buffers must be 16-byte aligned");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  ncg::MsmPlan pl;
  int prc = ncg_msm_plan_ws(ctx, curve, n, 0, &pl);
  if (prc) return prc;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  uint32_t bad = 0xFFFFFFFFu;
  NCG_HIP(ctx, ncg::msm_run(curve, pl, (const uint32_t*)points_affine_dev, (const uint32_t*)scalars_dev, ctx->msm_ws,
                            (uint32_t*)out_affine, &inf_local, st, &bad, &ctx->msm_side));
  if (bad != 0xFFFFFFFFu)  // validateMSMScalars (curve.ts:398-404): scalars must be below the group order
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: invalid scalar at index %u (not below the group order)", bad);
  if (out_is_inf) *out_is_inf = inf_local;
  return NCG_OK;
}

int ncg_msm(ncg_ctx* ctx, int curve, size_t n, const void* points_affine, const void* scalars, void* out_affine,
            uint8_t* out_is_inf) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  int pb = ncg_point_bytes(curve);
  if (pb == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: msm: unsupported curve %d", curve);
  if (n == 0) return ncg_msm_dev(ctx, curve, 0, nullptr, nullptr, out_affine, out_is_inf, nullptr);
  if (!points_affine || !scalars || !out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: too many points");
  PinSet pins(ctx);
  size_t pts_b = n * pb, sc_b = n * 32;
  const size_t pts_al = (pts_b + 255) & ~(size_t)255, sc_al = (sc_b + 255) & ~(size_t)255;
  const size_t stored_b = n * ncg::msm_stored_words_per_point(curve) * 4;
  int rc = ensure_scratch(ctx, pts_al + sc_al + stored_b + 2048);
  if (rc) return rc;
  char* d_pts = (char*)ctx->scratch;
  char* d_sc = d_pts + pts_al;
  if (n >= ((size_t)1 << 16)) {
    // Points AND scalars cross in PARTS, each part's scalars (a quarter of its bytes) ahead of its points: a part's digits and
    // counting sort need only its own scalars, its points are converted to the accumulate kernel's storage format on the side
    // stream as they land, and it is accumulated into the shared buckets (MsmPlan::part_flags) while the next part is still on
    // the bus - only the last part's accumulate, the fold and the tail run after the last byte.  2^20 G1 points from pinned host
    // memory: 9.3 ms (round 2, one copy then one MSM) -> 6.2 ms (sort under the transfer) -> 5.4 ms (parts; all scalars first)
    // -> see profiles/r04_host_path.json (the first part is ready after 32 MB instead of 56).
    rc = ensure_copy_streams(ctx);
    if (rc) return rc;
    char* d_stored = d_sc + sc_al;
    // measured on one box (tools/host_parts_sweep.py): G1 2^20 6.0 / 5.0 / 4.5 / 4.9 ms with 1 / 2 / 4 / 8 parts, G2 2^18 5.7 / 4.3 / 4.4 / 6.2
    int parts = n >= ((size_t)1 << 19) ? 4 : n >= ((size_t)1 << 17) ? 2 : 1;
    {
      const int k = ncg::knob("NCG_MSM_HOST_PARTS", 0);   // A/B builds only
      if (k >= 1 && k <= ncg_ctx::COPY_CHUNKS) parts = k;
    }
    const size_t per = (((n + parts - 1) / parts) + 255) & ~(size_t)255;
    ncg::MsmPlan whole, layout;
    if (ncg::msm_make_plan(curve, (int)n, 0, &whole) != 0 || ncg::msm_make_plan(curve, (int)std::min(n, per), whole.c, &layout) != 0)
      return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: cannot plan windows");
    layout.pts_stored = 1;
    layout.n_layout = layout.n;
    layout.Q_layout = layout.Q;
    rc = msm_ensure_ws(ctx, curve, layout);
    if (rc) return rc;
    // page locking per PART, right before the part's copies are enqueued: the copies are asynchronous, so the host locks the
    // pages of part p + 1 while part p is on the bus (locking all 128 MB of a 2^20-point G1 MSM up front costs ~3 ms before the
    // first byte moves).  Buffers the caller pinned with ncg_host_register are left alone (the registration attempt fails fast).
    const size_t sw = ncg::msm_stored_words_per_point(curve) * 4;
    hipError_t e = hipSuccess;
    PagedParts in_pts(points_affine, pts_b), in_sc(scalars, sc_b);
    const uint32_t *d_fin = nullptr, *d_bad = nullptr;
    ncg::MsmPlan last = layout;
    for (int p = 0; p < parts && e == hipSuccess; p++) {
      const size_t lo = std::min(n, per * (size_t)p), cnt = std::min(n, lo + per) - lo;
      const bool is_last = p == parts - 1 || lo + cnt >= n;
      if (cnt) {
        size_t a, b;
        if (in_sc.next((lo + cnt) * 32, is_last, true, &a, &b)) {
          pins.pin((const char*)scalars + a, b - a);
          e = hipMemcpyAsync(d_sc + a, (const char*)scalars + a, b - a, hipMemcpyHostToDevice, ctx->copy_in);
        }
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_sc[p], ctx->copy_in);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->ev_sc[p], 0);   // this part's digits wait for its scalars only
        if (e == hipSuccess && in_pts.next((lo + cnt) * pb, is_last, true, &a, &b)) {
          pins.pin((const char*)points_affine + a, b - a);
          e = hipMemcpyAsync(d_pts + a, (const char*)points_affine + a, b - a, hipMemcpyHostToDevice, ctx->copy_in);
        }
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_in[p], ctx->copy_in);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->msm_side.stream, ctx->ev_in[p], 0);
        if (e == hipSuccess)
          e = ncg::msm_points_to_stored(curve, (const uint32_t*)(d_pts + lo * pb), (int)cnt, (uint32_t*)(d_stored + lo * sw), ctx->msm_side.stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_k[p], ctx->msm_side.stream);
        if (e != hipSuccess) break;
      }
      ncg::MsmPlan pl;
      if (ncg::msm_make_plan(curve, (int)std::max<size_t>(cnt, 1), whole.c, &pl) != 0) {
        (void)drain_copy_streams(ctx);   // copies of this and earlier parts are in flight
        return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: cannot plan windows");
      }
      msm_apply_ctx(ctx, pl);
      pl.pts_stored = 1;
      pl.n_layout = layout.n;
      pl.Q_layout = layout.Q;
      pl.index_base = (uint32_t)lo;
      pl.part_flags = (p == 0 ? 1 : 0) | (is_last ? 2 : 0);
      ncg::MsmSide side;            // no fork / join of its own: the conversion is already in flight on the side stream
      side.pts_ready = cnt ? ctx->ev_k[p] : nullptr;
      e = ncg::msm_device_phase(curve, pl, (const uint32_t*)(d_stored + lo * sw), (const uint32_t*)(d_sc + lo * 32), ctx->msm_ws, &d_fin,
                                ctx->stream, &d_bad, &side);
      last = pl;
      if (is_last) break;
    }
    uint32_t bad = 0xFFFFFFFFu;
    uint8_t inf_local = 0;
    if (e == hipSuccess) e = ncg::msm_finish(curve, last, d_fin, (uint32_t*)out_affine, &inf_local, ctx->stream, d_bad, &bad);
    const hipError_t ed = drain_copy_streams(ctx);
    if (e == hipSuccess) e = ed;
    if (e != hipSuccess) return set_err(ctx, NCG_ERR_HIP, "noble-gpu: msm: %s", hipGetErrorString(e));
    if (bad != 0xFFFFFFFFu)
      return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: invalid scalar at index %u (not below the group order)", bad);
    if (out_is_inf) *out_is_inf = inf_local;
    return NCG_OK;
  }
  NCG_HIP(ctx, pins.h2d(d_pts, points_affine, pts_b));
  NCG_HIP(ctx, pins.h2d(d_sc, scalars, sc_b));
  return ncg_msm_dev(ctx, curve, n, d_pts, d_sc, out_affine, out_is_inf, ctx->stream);
}

// ---- resident point sets: upload once, multiply many (interleavedMSMUnsafe's usage pattern,
// src/abstract/curve.ts:907-959; SURVEY 8a gotcha 8: marshalling dominates an end-to-end call)

// The point of a resident set (curve.ts:907-918: precompute once, call with scalars): the wire -> storage
// conversion of the points is paid at the first MSM, every later call starts at the digits.
static int points_build_stored(ncg_ctx* ctx, ncg_points* h, hipStream_t st) {
  if (h->d_stored || h->n == 0) return NCG_OK;
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, h->n * ncg::msm_stored_words_per_point(h->curve) * 4);
  if (e != hipSuccess) {  // not fatal: the generic path converts per call
    (void)hipGetLastError();
    return NCG_OK;
  }
  e = ncg::msm_points_to_stored(h->curve, (const uint32_t*)h->d_pts, (int)h->n, (uint32_t*)d, st);
  // the cache is published only once the conversion has FINISHED: another lane's stream (ncg_msm_async_submit) or a later call on
  // ctx->stream reads d_stored with no ordering against `st` otherwise (one synchronisation per set, as points_build_endo does)
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return set_err(ctx, NCG_ERR_HIP, "noble-gpu: points_to_stored: %s", hipGetErrorString(e));
  }
  h->d_stored = d;
  return NCG_OK;
}

// Build the endomorphism images of a set whose points are KNOWN to lie in the prime-order subgroup.
static int points_build_endo(ncg_ctx* ctx, ncg_points* h) {
  const int E = ncg::msm_endo_factor(h->curve);
  if (E == 0 || h->n == 0 || h->d_endo) return NCG_OK;
  if (h->n * (size_t)E > 0x7fffffffu) return NCG_OK;  // too large for the expanded index space: generic path
  const size_t bytes = h->n * (size_t)E * ncg::msm_endo_words_per_point(h->curve) * 4;
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, bytes);
  if (e != hipSuccess) return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  e = ncg::msm_endo_expand(h->curve, (const uint32_t*)h->d_pts, (int)h->n, (uint32_t*)d, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return set_err(ctx, NCG_ERR_HIP, "noble-gpu: endomorphism images: %s", hipGetErrorString(e));
  }
  h->d_endo = d;
  return NCG_OK;
}

int ncg_points_upload(ncg_ctx* ctx, int curve, size_t n, const void* points_affine, ncg_points** out) {
  if (!ctx || !out) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: points_upload: NULL argument");
  *out = nullptr;
  const int pb = ncg_point_bytes(curve);
  if (pb == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: points_upload: unsupported curve %d", curve);
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (n && !points_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: points_upload: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  ncg_points* h = new ncg_points{ctx, curve, n, nullptr};
  if (n) {
    hipError_t e = hipMalloc(&h->d_pts, n * (size_t)pb);
    if (e != hipSuccess) {
      delete h;
      return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipMalloc(%zu) failed: %s", n * (size_t)pb, hipGetErrorString(e));
    }
    PinSet pins(ctx);
    e = pins.h2d(h->d_pts, points_affine, n * (size_t)pb);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      (void)hipFree(h->d_pts);
      delete h;
      return set_err(ctx, NCG_ERR_HIP, "noble-gpu: points_upload: copy failed: %s", hipGetErrorString(e));
    }
  }
  *out = h;
  return NCG_OK;
}

int ncg_points_from_encoded(ncg_ctx* ctx, int curve, size_t n, const void* encoded, int flags, ncg_points** out,
                            int64_t* out_bad_index) {
  if (!ctx || !out) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: points_from_encoded: NULL argument");
  *out = nullptr;
  if (out_bad_index) *out_bad_index = -1;
  const int ib = ncg::decode_in_bytes(curve), pb = ncg_point_bytes(curve);
  if (ib == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: points_from_encoded: unsupported curve %d", curve);
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (n && !encoded) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: points_from_encoded: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  ncg_points* h = new ncg_points{ctx, curve, n, nullptr};
  if (n) {
    hipError_t e = hipMalloc(&h->d_pts, n * (size_t)pb);
    if (e != hipSuccess) {
      delete h;
      return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipMalloc(%zu) failed: %s", n * (size_t)pb, hipGetErrorString(e));
    }
    std::vector<uint8_t> ok(n);
    const size_t in_b = (n * (size_t)ib + 255) & ~(size_t)255, fl_b = (n + 255) & ~(size_t)255;
    int rc = ensure_scratch(ctx, in_b + 2 * fl_b + 1024);
    if (rc == NCG_OK) {
      PinSet pins(ctx);
      char* d_in = (char*)ctx->scratch;
      char* d_ok = d_in + in_b;
      char* d_inf = d_ok + fl_b;
      hipError_t e2 = pins.h2d(d_in, encoded, n * (size_t)ib);
      if (e2 == hipSuccess)
        rc = ncg_decode_points_batch_dev(ctx, curve, n, d_in, flags, h->d_pts, (uint8_t*)d_ok, (uint8_t*)d_inf, ctx->stream);
      if (e2 == hipSuccess && rc == NCG_OK) e2 = pins.d2h(ok.data(), d_ok, n);
      if (e2 == hipSuccess && rc == NCG_OK) e2 = hipStreamSynchronize(ctx->stream);
      if (e2 != hipSuccess) rc = set_err(ctx, NCG_ERR_HIP, "noble-gpu: points_from_encoded: %s", hipGetErrorString(e2));
    }
    if (rc == NCG_OK)
      for (size_t i = 0; i < n; i++)
        if (!ok[i]) {  // the reference throws while decoding (Point.fromBytes / assertValidity)
          if (out_bad_index) *out_bad_index = (int64_t)i;
          rc = set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: points_from_encoded: invalid point encoding at index %zu", i);
          break;
        }
    // the bls12-381 decoders include the subgroup test (bls12-381.ts:567-577, :599-601), so a decoded set
    // qualifies for the endomorphism MSM; infinity encodings decode to ZERO and stay ZERO in every image
    if (rc == NCG_OK) (void)points_build_endo(ctx, h);  // best effort: without the images the set uses the generic MSM
    if (rc != NCG_OK) {
      (void)hipFree(h->d_pts);
      delete h;
      return rc;
    }
  }
  *out = h;
  return NCG_OK;
}

int ncg_points_verify_subgroup(ncg_ctx* ctx, ncg_points* h, int64_t* out_bad_index) {
  if (out_bad_index) *out_bad_index = -1;
  if (!ctx || !h || h->ctx != ctx) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: points_verify_subgroup: handle does not belong to this context");
  const int E = ncg::msm_endo_factor(h->curve);
  if (E == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: points_verify_subgroup: bls12-381 G1 / G2 only");
  if (h->n == 0 || h->d_endo) return NCG_OK;
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = points_build_endo(ctx, h);
  if (rc != NCG_OK || !h->d_endo) return rc;
  // [z^2] P (G1) / [z] P (G2) by the generic (complete) batch multiply, compared with the first image
  const size_t n = h->n, pb = (size_t)ncg_point_bytes(h->curve);
  char* tmp = nullptr;
  const size_t sc_b = (n * 32 + 255) & ~(size_t)255, out_b = (n * pb + 255) & ~(size_t)255, inf_b = (n + 255) & ~(size_t)255;
  hipError_t e = hipMalloc((void**)&tmp, sc_b + out_b + inf_b + 256);
  auto drop = [&]() {
    if (tmp) (void)hipFree(tmp);
    (void)hipFree(h->d_endo);
    h->d_endo = nullptr;
  };
  if (e != hipSuccess) {
    tmp = nullptr;
    drop();
    return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipMalloc failed: %s", hipGetErrorString(e));
  }
  uint32_t kz[8];
  ncg::msm_endo_verify_scalar(h->curve, kz);
  std::vector<uint32_t> sc(n * 8);
  for (size_t i = 0; i < n; i++) memcpy(&sc[i * 8], kz, 32);
  uint32_t bad = 0xFFFFFFFFu;
  uint32_t* d_bad = (uint32_t*)(tmp + sc_b + out_b + inf_b);
  e = hipMemcpyAsync(tmp, sc.data(), n * 32, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_bad, &bad, 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // `sc` and `bad` are pageable
  if (e == hipSuccess) {
    rc = ncg_mul_var_batch_dev(ctx, h->curve, n, h->d_pts, tmp, tmp + sc_b, (uint8_t*)(tmp + sc_b + out_b), ctx->stream);
    if (rc != NCG_OK) {
      drop();
      return rc;
    }
    e = ncg::msm_endo_verify(h->curve, (const uint32_t*)(tmp + sc_b), (const uint8_t*)(tmp + sc_b + out_b),
                             (const uint32_t*)h->d_endo, (int)n, d_bad, ctx->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    drop();
    return set_err(ctx, NCG_ERR_HIP, "noble-gpu: points_verify_subgroup: %s", hipGetErrorString(e));
  }
  (void)hipFree(tmp);
  tmp = nullptr;
  if (bad != 0xFFFFFFFFu) {  // not an error: the set stays usable through the generic path
    drop();
    if (out_bad_index) *out_bad_index = (int64_t)bad;
  }
  return NCG_OK;
}

// interleavedMSMUnsafe's precomputation (curve.ts:907-959: tables once per point set) on the device: window-shifted
// copies of the set, after which ncg_msm_resident* runs the shared-bucket MSM (one bucket fold, no Horner across
// windows).  Uses the endomorphism images when the set is verified, the points themselves otherwise.  Sets below
// 4096 points and ed25519 sets are left as they are (NCG_OK, nothing built).
int ncg_points_precompute(ncg_ctx* ctx, ncg_points* h) {
  if (!ctx || !h || h->ctx != ctx) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: points_precompute: handle does not belong to this context");
  if (h->d_shift || h->n < 4096 || h->curve == NCG_ED25519) return NCG_OK;
  static const bool no_shift = ncg::knob_set("NCG_NO_PRECOMP");
  if (no_shift) return NCG_OK;  // the levels would never be used: pay neither the build nor the memory
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  static const bool no_endo = ncg::knob_set("NCG_NO_ENDO");
  const bool endo = h->d_endo && !no_endo;
  ncg::MsmPlan pl;
  const int c = 16;
  if ((endo ? ncg::msm_make_plan_endo(h->curve, (int)h->n, c, &pl) : ncg::msm_make_plan(h->curve, (int)h->n, c, &pl)) != 0)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: cannot plan windows");
  const size_t m = (size_t)pl.n, sw = ncg::msm_stored_words_per_point(h->curve);
  if (m * (size_t)pl.nwin > 0x7fffffffu) return NCG_OK;  // entry index space: keep the per-window path
  if (!endo) {
    int rc = points_build_stored(ctx, h, ctx->stream);
    if (rc) return rc;
    if (!h->d_stored) return NCG_OK;
  }
  void *d = nullptr, *tmp = nullptr;
  {  // the levels are nwin copies of the set: leave room for the MSM workspace and whatever else the process allocates
    size_t free_b = 0, total_b = 0;
    const size_t want = m * (size_t)pl.nwin * sw * 4 + ncg::msm_shift_tmp_bytes(h->curve, (int)m);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b / 2) return NCG_OK;  // keep the per-window path
  }
  hipError_t e = hipMalloc(&d, m * (size_t)pl.nwin * sw * 4);
  if (e == hipSuccess) e = hipMalloc(&tmp, ncg::msm_shift_tmp_bytes(h->curve, (int)m));
  if (e != hipSuccess) {  // not fatal: the set keeps the per-window path
    (void)hipGetLastError();
    if (d) (void)hipFree(d);
    return NCG_OK;
  }
  e = hipMemcpyAsync(d, endo ? h->d_endo : h->d_stored, m * sw * 4, hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = ncg::msm_shift_levels(h->curve, (uint32_t*)d, (int)m, pl.nwin, pl.c, tmp, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return set_err(ctx, NCG_ERR_HIP, "noble-gpu: points_precompute: %s", hipGetErrorString(e));
  }
  if (!endo && h->d_stored) {  // level 0 of the copies IS the stored set: the separate array is a duplicate now
    (void)hipFree(h->d_stored);
    h->d_stored = nullptr;
  }
  h->d_shift = d;
  h->shift_c = pl.c;
  h->shift_nwin = pl.nwin;
  h->shift_mode = endo ? 2 : 1;
  h->shift_m = m;
  return NCG_OK;
}
int ncg_points_precomputed(const ncg_points* h) { return h && h->d_shift ? 1 : 0; }

int ncg_points_in_subgroup(const ncg_points* h) { return h && h->d_endo ? 1 : 0; }

void ncg_points_free(ncg_points* h) {
  if (!h) return;
  (void)hipSetDevice(h->ctx->device);
  if (h->d_pts) (void)hipFree(h->d_pts);
  if (h->d_endo) (void)hipFree(h->d_endo);
  if (h->d_stored) (void)hipFree(h->d_stored);
  if (h->d_shift) (void)hipFree(h->d_shift);
  delete h;
}
size_t ncg_points_count(const ncg_points* h) { return h ? h->n : 0; }
int ncg_points_curve(const ncg_points* h) { return h ? h->curve : -1; }
const void* ncg_points_dev(const ncg_points* h) { return h ? h->d_pts : nullptr; }

// scalars -> the tail of the scratch buffer; returns the device address
static int upload_scalars(ncg_ctx* ctx, PinSet& pins, size_t n, const void* scalars, char** d_sc) {
  int rc = ensure_scratch(ctx, n * 32 + 512);
  if (rc) return rc;
  *d_sc = (char*)ctx->scratch;
  NCG_HIP(ctx, pins.h2d(*d_sc, scalars, n * 32));
  return NCG_OK;
}

}  // extern "C"
#pragma GCC visibility pop
// Which window plan and which device point array an MSM on a resident set uses: the precomputed levels (shared-bucket
// mode), the endomorphism images of a verified set, the stored form of the points, or - if that cache could not be
// allocated - the wire points.  The stored form is built on first use (on `st`).
int ncg_resident_plan(ncg_ctx* ctx, const ncg_points* pts, ncg::MsmPlan* pl, const uint32_t** d_pts, hipStream_t st) {
  static const bool no_endo = ncg::knob_set("NCG_NO_ENDO");
  static const bool no_shift = ncg::knob_set("NCG_NO_PRECOMP");
  if (pts->d_shift && !no_shift) {  // precomputed set: every window adds into one bucket set (msm.hpp `shared`)
    const int prc = pts->shift_mode == 2 ? ncg::msm_make_plan_endo(pts->curve, (int)pts->n, pts->shift_c, pl)
                                         : ncg::msm_make_plan(pts->curve, (int)pts->n, pts->shift_c, pl);
    if (prc != 0 || pl->nwin != pts->shift_nwin || (size_t)pl->n != pts->shift_m)
      return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: the precomputed levels do not match the window plan");
    pl->shared = 1;
    pl->top_tb = 0;       // one bucket set for all windows: the weight of a bucket is its index, no spreading
    pl->top_submask = 0;
    pl->pts_stored = 1;
    *d_pts = (const uint32_t*)pts->d_shift;
    return NCG_OK;
  }
  if (pts->d_endo && !no_endo) {  // verified subgroup set: endomorphism MSM on the expanded images (endo.hpp)
    if (ncg::msm_make_plan_endo(pts->curve, (int)pts->n, 0, pl) != 0)
      return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: cannot plan windows");
    *d_pts = (const uint32_t*)pts->d_endo;
    return NCG_OK;
  }
  if (ncg::msm_make_plan(pts->curve, (int)pts->n, 0, pl) != 0)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: cannot plan windows");
  if (pts->d_shift && pts->shift_mode == 1) {  // level 0 of the copies is the stored set
    pl->pts_stored = 1;
    *d_pts = (const uint32_t*)pts->d_shift;
    return NCG_OK;
  }
  int rc = points_build_stored(ctx, const_cast<ncg_points*>(pts), st);  // a cache inside the handle
  if (rc) return rc;
  if (pts->d_stored) {
    pl->pts_stored = 1;
    *d_pts = (const uint32_t*)pts->d_stored;
  } else {
    *d_pts = (const uint32_t*)pts->d_pts;  // wire points: converted per call
  }
  return NCG_OK;
}

#pragma GCC visibility push(default)
extern "C" {

// MSM on a resident set with the scalars already on the device
static int msm_resident_core(ncg_ctx* ctx, const ncg_points* pts, const void* d_sc, void* out_affine, uint8_t* out_is_inf,
                             hipStream_t st) {
  ncg::MsmPlan pl;
  const uint32_t* d_pts = nullptr;
  int rc = ncg_resident_plan(ctx, pts, &pl, &d_pts, st);
  if (rc) return rc;
  rc = msm_ensure_ws(ctx, pts->curve, pl);
  if (rc) return rc;
  uint32_t bad = 0xFFFFFFFFu;
  uint8_t inf_local = 0;
  NCG_HIP(ctx, ncg::msm_run(pts->curve, pl, d_pts, (const uint32_t*)d_sc, ctx->msm_ws, (uint32_t*)out_affine, &inf_local, st, &bad,
                            pl.pts_stored || pl.endo ? nullptr : &ctx->msm_side));
  if (bad != 0xFFFFFFFFu)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: invalid scalar at index %u (not below the group order)", bad);
  if (out_is_inf) *out_is_inf = inf_local;
  return NCG_OK;
}

int ncg_msm_resident(ncg_ctx* ctx, const ncg_points* pts, const void* scalars, void* out_affine, uint8_t* out_is_inf) {
  if (!ctx || !pts || pts->ctx != ctx) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_resident: handle does not belong to this context");
  if (pts->n == 0) return ncg_msm_dev(ctx, pts->curve, 0, nullptr, nullptr, out_affine, out_is_inf, nullptr);
  if (!scalars || !out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_resident: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  char* d_sc = nullptr;
  int rc = upload_scalars(ctx, pins, pts->n, scalars, &d_sc);
  if (rc) return rc;
  return msm_resident_core(ctx, pts, d_sc, out_affine, out_is_inf, ctx->stream);
}

int ncg_msm_resident_dev(ncg_ctx* ctx, const ncg_points* pts, const void* scalars_dev, void* out_affine, uint8_t* out_is_inf,
                         void* stream) {
  if (!ctx || !pts || pts->ctx != ctx) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_resident: handle does not belong to this context");
  if (pts->n == 0) return ncg_msm_dev(ctx, pts->curve, 0, nullptr, nullptr, out_affine, out_is_inf, nullptr);
  if (!scalars_dev || !out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_resident: NULL buffer");
  if (misaligned16(scalars_dev)) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_resident: device buffers must be 16-byte aligned");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  return msm_resident_core(ctx, pts, scalars_dev, out_affine, out_is_inf, stream ? (hipStream_t)stream : ctx->stream);
}

// batch multiply on a resident set, device buffers: a verified subgroup set takes the endomorphism ladders of
// mulvar_endo.hip (G1: two 128-bit streams along phi; G2: four 64-bit streams along psi), any other set the generic kernel
static int mul_var_resident_core(ncg_ctx* ctx, const ncg_points* pts, const void* d_sc, void* d_out, uint8_t* d_inf, hipStream_t st) {
  const size_t n = pts->n;
  static const bool no_endo = ncg::knob_set("NCG_NO_ENDO");
  const bool bls = pts->curve == NCG_BLS12_381_G1 || pts->curve == NCG_BLS12_381_G2;
  if (pts->d_endo && bls && !no_endo) {
    int rc = ensure_mul_ws(ctx, pts->curve, n, st);
    if (rc) return rc;
    if (pts->curve == NCG_BLS12_381_G1)
      NCG_HIP(ctx, ncg::mul_var_batch_g1_subgroup((const uint32_t*)pts->d_pts, (const uint32_t*)d_sc, (uint32_t*)d_out, d_inf, (int)n,
                                                  (uint32_t*)ctx->mul_ws, st));
    else
      NCG_HIP(ctx, ncg::mul_var_batch_g2_subgroup((const uint32_t*)pts->d_pts, (const uint32_t*)d_sc, (uint32_t*)d_out, d_inf, (int)n,
                                                  (uint32_t*)ctx->mul_ws, st));
    return NCG_OK;
  }
  return ncg_mul_var_batch_dev(ctx, pts->curve, n, pts->d_pts, d_sc, d_out, d_inf, st);
}

int ncg_mul_var_batch_resident_dev(ncg_ctx* ctx, const ncg_points* pts, const void* scalars_dev, void* out_affine_dev,
                                   uint8_t* out_is_inf_dev, void* stream) {
  if (!ctx || !pts || pts->ctx != ctx) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: mul_var_batch_resident: handle does not belong to this context");
  if (pts->n == 0) return NCG_OK;
  if (!scalars_dev || !out_affine_dev || !out_is_inf_dev) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: mul_var_batch_resident: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  return mul_var_resident_core(ctx, pts, scalars_dev, out_affine_dev, out_is_inf_dev, stream ? (hipStream_t)stream : ctx->stream);
}

int ncg_mul_var_batch_resident(ncg_ctx* ctx, const ncg_points* pts, const void* scalars, void* out_affine,
                               uint8_t* out_is_inf) {
  if (!ctx || !pts || pts->ctx != ctx) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: mul_var_batch_resident: handle does not belong to this context");
  const size_t n = pts->n;
  if (n == 0) return NCG_OK;
  if (!scalars || !out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: mul_var_batch_resident: NULL buffer");
  const int pb = ncg_point_bytes(pts->curve);
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  const size_t sc_b = (n * 32 + 255) & ~(size_t)255, out_b = (n * (size_t)pb + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, sc_b + out_b + n + 1024);
  if (rc) return rc;
  char* d_sc = (char*)ctx->scratch;
  char* d_out = d_sc + sc_b;
  char* d_inf = d_out + out_b;
  NCG_HIP(ctx, pins.h2d(d_sc, scalars, n * 32));
  rc = mul_var_resident_core(ctx, pts, d_sc, d_out, (uint8_t*)d_inf, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_affine, d_out, n * (size_t)pb));
  std::vector<uint8_t> inf_tmp;
  uint8_t* inf_dst = out_is_inf;
  if (!inf_dst) {
    inf_tmp.resize(n);
    inf_dst = inf_tmp.data();
  }
  NCG_HIP(ctx, pins.d2h(inf_dst, d_inf, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

// decode on the device, then sum through the MSM path with unit scalars; nothing but the encodings
// goes up and one point (plus the verdicts) comes back
int ncg_aggregate_encoded(ncg_ctx* ctx, int curve, size_t n, const void* encoded, int flags, void* out_affine,
                          uint8_t* out_is_inf, int64_t* out_bad_index) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  const int ib = ncg::decode_in_bytes(curve), pb = ncg_point_bytes(curve);
  if (ib == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: aggregate_encoded: unsupported curve %d", curve);
  if (out_bad_index) *out_bad_index = -1;
  if (n == 0) return ncg_msm_dev(ctx, curve, 0, nullptr, nullptr, out_affine, out_is_inf, nullptr);
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!encoded || !out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: aggregate_encoded: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint8_t> ok(n);
  const size_t in_b = (n * (size_t)ib + 255) & ~(size_t)255, pts_b = (n * (size_t)pb + 255) & ~(size_t)255,
               sc_b = n * 32, fl_b = (n + 255) & ~(size_t)255;
  {
    PinSet pins(ctx);
    int rc = ensure_scratch(ctx, in_b + pts_b + sc_b + 2 * fl_b + 2048);
    if (rc) return rc;
    char* d_in = (char*)ctx->scratch;
    char* d_pts = d_in + in_b;
    char* d_sc = d_pts + pts_b;
    char* d_ok = d_sc + sc_b;
    char* d_inf = d_ok + fl_b;
    NCG_HIP(ctx, pins.h2d(d_in, encoded, n * (size_t)ib));
    rc = ncg_decode_points_batch_dev(ctx, curve, n, d_in, flags, d_pts, (uint8_t*)d_ok, (uint8_t*)d_inf, ctx->stream);
    if (rc) return rc;
    NCG_HIP(ctx, hipMemsetAsync(d_sc, 0, sc_b, ctx->stream));
    NCG_HIP(ctx, hipMemset2DAsync(d_sc, 32, 1, 1, n, ctx->stream));  // scalar 1 in every 32-byte row
    NCG_HIP(ctx, pins.d2h(ok.data(), d_ok, n));
    NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  for (size_t i = 0; i < n; i++)
    if (!ok[i]) {  // the reference throws while decoding (Point.fromBytes / assertValidity)
      if (out_bad_index) *out_bad_index = (int64_t)i;
      return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: aggregate_encoded: invalid point encoding at index %zu", i);
    }
  char* d_pts = (char*)ctx->scratch + in_b;
  char* d_sc = d_pts + pts_b;
  return ncg_msm_dev(ctx, curve, n, d_pts, d_sc, out_affine, out_is_inf, ctx->stream);
}

int ncg_normalize_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* points_proj_dev, void* out_affine_dev,
                            uint8_t* out_is_inf_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (ncg_point_bytes(curve) == 0)
    return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: normalize_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!points_proj_dev || !out_affine_dev || !out_is_inf_dev)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: normalize_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  NCG_HIP(ctx, ncg::normalize_batch(curve, (const uint32_t*)points_proj_dev, (uint32_t*)out_affine_dev, out_is_inf_dev,
                                    (int)n, st));
  return NCG_OK;
}

int ncg_normalize_batch(ncg_ctx* ctx, int curve, size_t n, const void* points_proj, void* out_affine,
                        uint8_t* out_is_inf) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  int pb = ncg_point_bytes(curve);
  if (pb == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: normalize_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (!points_proj || !out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: normalize_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  size_t in_b = n * (size_t)(pb / 2) * 3, out_b = n * (size_t)pb, inf_b = (n + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, in_b + out_b + inf_b + 2048);
  if (rc) return rc;
  char* d_in = (char*)ctx->scratch;
  char* d_out = d_in + ((in_b + 255) & ~(size_t)255);
  char* d_inf = d_out + out_b;
  NCG_HIP(ctx, pins.h2d(d_in, points_proj, in_b));
  rc = ncg_normalize_batch_dev(ctx, curve, n, d_in, d_out, (uint8_t*)d_inf, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_affine, d_out, out_b));
  if (out_is_inf) NCG_HIP(ctx, pins.d2h(out_is_inf, d_inf, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

int ncg_decode_points_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* encoded_dev, int flags,
                                void* out_affine_dev, uint8_t* out_ok_dev, uint8_t* out_is_inf_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (ncg::decode_in_bytes(curve) == 0)
    return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: decode_points_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!encoded_dev || !out_affine_dev || !out_ok_dev || !out_is_inf_dev)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: decode_points_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  NCG_HIP(ctx, ncg::decode_points_batch(curve, (const uint8_t*)encoded_dev, flags, (uint32_t*)out_affine_dev, out_ok_dev,
                                        out_is_inf_dev, (int)n, st));
  return NCG_OK;
}

int ncg_decode_points_batch(ncg_ctx* ctx, int curve, size_t n, const void* encoded, int flags, void* out_affine,
                            uint8_t* out_ok, uint8_t* out_is_inf) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  int ib = ncg::decode_in_bytes(curve), pb = ncg_point_bytes(curve);
  if (ib == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: decode_points_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (!encoded || !out_affine || !out_ok)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: decode_points_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  size_t in_b = (n * ib + 255) & ~(size_t)255, out_b = n * (size_t)pb, fl_b = (n + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, in_b + out_b + 2 * fl_b + 1024);
  if (rc) return rc;
  char* d_in = (char*)ctx->scratch;
  char* d_out = d_in + in_b;
  char* d_ok = d_out + out_b;
  char* d_inf = d_ok + fl_b;
  NCG_HIP(ctx, pins.h2d(d_in, encoded, n * ib));
  rc = ncg_decode_points_batch_dev(ctx, curve, n, d_in, flags, d_out, (uint8_t*)d_ok, (uint8_t*)d_inf, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_affine, d_out, out_b));
  NCG_HIP(ctx, pins.d2h(out_ok, d_ok, n));
  if (out_is_inf) NCG_HIP(ctx, pins.d2h(out_is_inf, d_inf, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

int ncg_encode_points_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* affine_dev, void* out_encoded_dev,
                                uint8_t* out_ok_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (ncg::decode_in_bytes(curve) == 0)
    return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: encode_points_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!affine_dev || !out_encoded_dev || !out_ok_dev)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: encode_points_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  NCG_HIP(ctx, ncg::encode_points_batch(curve, (const uint32_t*)affine_dev, (uint8_t*)out_encoded_dev, out_ok_dev, (int)n, st));
  return NCG_OK;
}

int ncg_encode_points_batch(ncg_ctx* ctx, int curve, size_t n, const void* affine, void* out_encoded, uint8_t* out_ok) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  int ob = ncg::decode_in_bytes(curve), pb = ncg_point_bytes(curve);
  if (ob == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: encode_points_batch: unsupported curve %d", curve);
  if (n == 0) return NCG_OK;
  if (!affine || !out_encoded || !out_ok)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: encode_points_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  size_t in_b = (n * (size_t)pb + 255) & ~(size_t)255, out_b = (n * (size_t)ob + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, in_b + out_b + n + 1024);
  if (rc) return rc;
  char* d_in = (char*)ctx->scratch;
  char* d_out = d_in + in_b;
  char* d_ok = d_out + out_b;
  NCG_HIP(ctx, pins.h2d(d_in, affine, n * (size_t)pb));
  rc = ncg_encode_points_batch_dev(ctx, curve, n, d_in, d_out, (uint8_t*)d_ok, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_encoded, d_out, n * (size_t)ob));
  NCG_HIP(ctx, pins.d2h(out_ok, d_ok, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

int ncg_map_to_curve_batch_dev(ncg_ctx* ctx, int curve, size_t n, int count, const void* u_dev, void* out_affine_dev,
                               uint8_t* out_is_inf_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (curve != NCG_BLS12_381_G1 && curve != NCG_BLS12_381_G2)
    return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: map_to_curve_batch: unsupported curve %d", curve);
  if (count != 1 && count != 2) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: map_to_curve_batch: count must be 1 or 2");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!u_dev || !out_affine_dev || !out_is_inf_dev)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: map_to_curve_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  int rc = ensure_mul_ws(ctx, curve, n, st);  // Jacobian scratch for the batched affine conversion
  if (rc) return rc;
  NCG_HIP(ctx, ncg::map_to_curve_batch(curve, (const uint32_t*)u_dev, count, (uint32_t*)out_affine_dev, out_is_inf_dev,
                                       (int)n, (uint32_t*)ctx->mul_ws, st));
  return NCG_OK;
}

int ncg_map_to_curve_batch(ncg_ctx* ctx, int curve, size_t n, int count, const void* u, void* out_affine,
                           uint8_t* out_is_inf) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (curve != NCG_BLS12_381_G1 && curve != NCG_BLS12_381_G2)
    return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: map_to_curve_batch: unsupported curve %d", curve);
  if (count != 1 && count != 2) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: map_to_curve_batch: count must be 1 or 2");
  if (n == 0) return NCG_OK;
  if (!u || !out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: map_to_curve_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  const int pb = ncg_point_bytes(curve);
  const size_t in_b = (n * (size_t)count * (pb / 2) + 255) & ~(size_t)255, out_b = n * (size_t)pb;
  int rc = ensure_scratch(ctx, in_b + out_b + n + 1024);
  if (rc) return rc;
  char* d_in = (char*)ctx->scratch;
  char* d_out = d_in + in_b;
  char* d_inf = d_out + out_b;
  NCG_HIP(ctx, pins.h2d(d_in, u, n * (size_t)count * (pb / 2)));
  rc = ncg_map_to_curve_batch_dev(ctx, curve, n, count, d_in, d_out, (uint8_t*)d_inf, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_affine, d_out, out_b));
  if (out_is_inf) NCG_HIP(ctx, pins.d2h(out_is_inf, d_inf, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

// twiddle table for (log2n, omega): built on first use, rebuilt if a different root is passed
static int ensure_ntt_table(ncg_ctx* ctx, int log2n, const uint32_t* omega) {
  if (ctx->ntt_tab[log2n] && memcmp(ctx->ntt_omega[log2n], omega, 32) == 0) return NCG_OK;
  if (ctx->ntt_tab[log2n]) {
    NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->ntt_tab[log2n]);
    ctx->ntt_tab[log2n] = nullptr;
  }
  uint32_t* tab = nullptr;
  void* tmp = nullptr;
  hipError_t e = hipMalloc((void**)&tab, ncg::ntt_table_bytes(log2n));
  if (e != hipSuccess) return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: ntt table hipMalloc failed: %s", hipGetErrorString(e));
  e = hipMalloc(&tmp, ncg::ntt_small_bytes(log2n) + 32);
  if (e != hipSuccess) {
    (void)hipFree(tab);
    return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: ntt table hipMalloc failed: %s", hipGetErrorString(e));
  }
  uint32_t* d_omega = (uint32_t*)tmp;
  uint32_t* d_small = d_omega + 8;
  alignas(16) uint32_t probe[16] = {0}, expect_tw[16] = {0};
  const int tww = ncg::ntt_tw_words();
  e = hipMemcpyAsync(d_omega, omega, 32, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = ncg::ntt_build_table(log2n, d_omega, d_small, tab, ctx->stream);
  // primitive-root check: omega^(N/2) == -1 (N = 1: omega == 1); table entries are x 2^261 mod r
  const size_t probe_idx = log2n ? ((size_t)1 << (log2n - 1)) : 0;
  if (e == hipSuccess) e = hipMemcpyAsync(probe, tab + probe_idx * tww, (size_t)tww * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) {
    (void)hipFree(tab);
    return set_err(ctx, NCG_ERR_HIP, "noble-gpu: ntt table build failed: %s", hipGetErrorString(e));
  }
  uint32_t expect[8];  // entries are x 2^261 mod r (ntt.hip): K261 for +1, r - K261 for -1
  if (log2n == 0) {
    for (int i = 0; i < 8; i++) expect[i] = ncg::Fr29PR::K261[i];
  } else {
    uint64_t bw = 0;
    for (int i = 0; i < 8; i++) {
      uint64_t d = (uint64_t)ncg::ParamsBlsR::P[i] - ncg::Fr29PR::K261[i] - bw;
      expect[i] = (uint32_t)d;
      bw = (d >> 32) & 1;
    }
  }
  ncg::ntt_tw_from_canonical(expect, expect_tw);
  if (memcmp(probe, expect_tw, (size_t)tww * 4) != 0) {
    (void)hipFree(tab);
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ntt: omega is not a primitive 2^%d-th root of unity", log2n);
  }
  ctx->ntt_tab[log2n] = tab;
  memcpy(ctx->ntt_omega[log2n], omega, 32);
  return NCG_OK;
}

int ncg_ntt_dev(ncg_ctx* ctx, int field, int log2n, size_t batch, const void* omega, const void* in_dev, void* out_dev,
                int flags, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (field != NCG_FIELD_BLS12_381_FR) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: ntt: unsupported field %d", field);
  if (log2n < 0 || log2n > NCG_NTT_MAX_LOG2N)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ntt: log2n %d out of range 0..%d", log2n, NCG_NTT_MAX_LOG2N);
  if (batch == 0) return NCG_OK;
  if (batch > 65535) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ntt: batch too large");
  if (!omega || !in_dev || !out_dev) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ntt: NULL buffer");
  if (misaligned16(in_dev) || misaligned16(out_dev)) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ntt: device buffers must be 16-byte aligned");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_ntt_table(ctx, log2n, (const uint32_t*)omega);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const bool fold = ((flags >> 1) & 1) == ((flags >> 2) & 1);
  const size_t bytes = (batch << log2n) * 32;
  if (fold && log2n > 10 && ctx->ntt_ws_bytes < bytes) {
    if (ctx->ntt_ws) {
      NCG_HIP(ctx, hipDeviceSynchronize());
      (void)hipFree(ctx->ntt_ws);
      ctx->ntt_ws = nullptr;
      ctx->ntt_ws_bytes = 0;
    }
    hipError_t e = hipMalloc(&ctx->ntt_ws, bytes);
    if (e != hipSuccess) return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: ntt workspace hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    ctx->ntt_ws_bytes = bytes;
  }
  NCG_HIP(ctx, ncg::ntt_run(log2n, batch, (const uint32_t*)in_dev, (uint32_t*)out_dev, (uint32_t*)ctx->ntt_ws,
                            ctx->ntt_tab[log2n], log2n, flags, st));
  return NCG_OK;
}

int ncg_ntt(ncg_ctx* ctx, int field, int log2n, size_t batch, const void* omega, const void* in, void* out, int flags) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (log2n < 0 || log2n > NCG_NTT_MAX_LOG2N)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ntt: log2n %d out of range 0..%d", log2n, NCG_NTT_MAX_LOG2N);
  if (field != NCG_FIELD_BLS12_381_FR) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: ntt: unsupported field %d", field);
  if (batch == 0) return NCG_OK;
  if (batch > 65535) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ntt: batch %zu too large (max 65535)", batch);
  if (!omega || !in || !out) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ntt: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  const size_t bytes = (batch << log2n) * 32;  // batch <= 2^16, log2n <= 28: below 2^49
  int rc = ensure_scratch(ctx, bytes + 1024);
  if (rc) return rc;
  NCG_HIP(ctx, pins.h2d(ctx->scratch, in, bytes));
  rc = ncg_ntt_dev(ctx, field, log2n, batch, omega, ctx->scratch, ctx->scratch, flags, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out, ctx->scratch, bytes));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

static int ensure_ed_table(ncg_ctx* ctx) {
  if (ctx->ed_btab) return NCG_OK;
  uint32_t host[ncg::ED25519_BTAB_WORDS];
  ncg::ed25519_build_base_table(host);
  NCG_HIP(ctx, hipMalloc((void**)&ctx->ed_btab, sizeof host));
  NCG_HIP(ctx, hipMemcpy(ctx->ed_btab, host, sizeof host, hipMemcpyHostToDevice));
  return NCG_OK;
}

int ncg_ed25519_verify_batch_dev(ncg_ctx* ctx, size_t n, const void* sig64_dev, const void* pk32_dev,
                                 const void* k32_dev, int zip215, uint8_t* out_ok_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig64_dev || !pk32_dev || !k32_dev || !out_ok_dev)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ed25519_verify_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_ed_table(ctx);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  rc = ensure_mul_ws(ctx, NCG_ED25519, n, st);  // per-item window tables live in the multiply scratch
  if (rc) return rc;
  NCG_HIP(ctx, ncg::ed25519_verify_batch((const uint32_t*)sig64_dev, (const uint32_t*)pk32_dev,
                                         (const uint32_t*)k32_dev, ctx->ed_btab, zip215, out_ok_dev, (int)n,
                                         (uint32_t*)ctx->mul_ws, st));
  return NCG_OK;
}

int ncg_ed25519_verify_batch(ncg_ctx* ctx, size_t n, const void* sig64, const void* pk32, const void* k32,
                             int zip215, uint8_t* out_ok) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (n == 0) return NCG_OK;
  if (!sig64 || !pk32 || !k32 || !out_ok)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ed25519_verify_batch: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  size_t okb = (n + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, n * 128 + okb + 1024);
  if (rc) return rc;
  char* d_sig = (char*)ctx->scratch;
  char* d_pk = d_sig + n * 64;
  char* d_k = d_pk + n * 32;
  char* d_ok = d_k + n * 32;
  NCG_HIP(ctx, pins.h2d(d_sig, sig64, n * 64));
  NCG_HIP(ctx, pins.h2d(d_pk, pk32, n * 32));
  NCG_HIP(ctx, pins.h2d(d_k, k32, n * 32));
  rc = ncg_ed25519_verify_batch_dev(ctx, n, d_sig, d_pk, d_k, zip215, (uint8_t*)d_ok, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_ok, d_ok, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

// ---- ed25519 verify from messages: the challenge hash runs on the device too
static int ensure_ed_ks(ncg_ctx* ctx, size_t n, hipStream_t st) {
  const size_t need = n * 32;
  if (ctx->ed_ks_bytes >= need) return NCG_OK;
  NCG_HIP(ctx, hipStreamSynchronize(st));
  if (ctx->ed_ks) (void)hipFree(ctx->ed_ks);
  ctx->ed_ks = nullptr;
  ctx->ed_ks_bytes = 0;
  hipError_t e = hipMalloc(&ctx->ed_ks, need + (need >> 2));
  if (e != hipSuccess) return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
  ctx->ed_ks_bytes = need + (need >> 2);
  return NCG_OK;
}

int ncg_ed25519_challenge_batch_dev(ncg_ctx* ctx, size_t n, const void* sig64_dev, const void* pk32_dev, const void* msgs_dev,
                                    const uint64_t* msg_off_dev, void* out_k32_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig64_dev || !pk32_dev || !msg_off_dev || !out_k32_dev)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ed25519_challenge: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  NCG_HIP(ctx, ncg::ed25519_challenge_batch((const uint8_t*)sig64_dev, (const uint8_t*)pk32_dev, (const uint8_t*)msgs_dev,
                                            msg_off_dev, (uint32_t*)out_k32_dev, (int)n, st));
  return NCG_OK;
}

int ncg_ed25519_verify_batch_msgs_dev(ncg_ctx* ctx, size_t n, const void* sig64_dev, const void* pk32_dev,
                                      const void* msgs_dev, const uint64_t* msg_off_dev, int zip215, uint8_t* out_ok_dev,
                                      void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!out_ok_dev) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ed25519_verify_msgs: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  int rc = ensure_ed_ks(ctx, n, st);
  if (rc) return rc;
  rc = ncg_ed25519_challenge_batch_dev(ctx, n, sig64_dev, pk32_dev, msgs_dev, msg_off_dev, ctx->ed_ks, st);
  if (rc) return rc;
  return ncg_ed25519_verify_batch_dev(ctx, n, sig64_dev, pk32_dev, ctx->ed_ks, zip215, out_ok_dev, st);
}

int ncg_ed25519_verify_batch_msgs(ncg_ctx* ctx, size_t n, const void* sig64, const void* pk32, const void* msgs,
                                  const uint64_t* msg_off, int zip215, uint8_t* out_ok) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig64 || !pk32 || !msg_off || !out_ok) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ed25519_verify_msgs: NULL buffer");
  for (size_t i = 0; i < n; i++)
    if (msg_off[i + 1] < msg_off[i]) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ed25519_verify_msgs: offsets must not decrease (index %zu)", i);
  const size_t mbytes = (size_t)(msg_off[n] - msg_off[0]);
  if (mbytes && !msgs) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ed25519_verify_msgs: NULL message buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  const size_t sig_b = n * 64, pk_b = n * 32, off_b = (n + 1) * 8;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  int rc = ensure_scratch(ctx, al(sig_b) + al(pk_b) + al(off_b) + al(mbytes + 8) + al(n) + 1024);
  if (rc) return rc;
  char* d_sig = (char*)ctx->scratch;
  char* d_pk = d_sig + al(sig_b);
  char* d_off = d_pk + al(pk_b);
  char* d_msg = d_off + al(off_b);
  char* d_ok = d_msg + al(mbytes + 8);
  NCG_HIP(ctx, pins.h2d(d_sig, sig64, sig_b));
  NCG_HIP(ctx, pins.h2d(d_pk, pk32, pk_b));
  std::vector<uint64_t> rel(n + 1);  // offsets relative to the first message
  for (size_t i = 0; i <= n; i++) rel[i] = msg_off[i] - msg_off[0];
  NCG_HIP(ctx, hipMemcpyAsync(d_off, rel.data(), off_b, hipMemcpyHostToDevice, ctx->stream));
  if (mbytes) NCG_HIP(ctx, pins.h2d(d_msg, (const char*)msgs + msg_off[0], mbytes));
  rc = ncg_ed25519_verify_batch_msgs_dev(ctx, n, d_sig, d_pk, d_msg, (const uint64_t*)d_off, zip215, (uint8_t*)d_ok,
                                         ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_ok, d_ok, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

// ---- secp256k1 ECDSA batch verify (weierstrass.ts:1571-1620): SEC1 decode of the keys, the scalar side
// (ecdsa.hip), u1 G by the fixed-base table, u2 P by the variable-base ladder, one pairwise add, compare.
struct SigWs {  // device buffers of the signature pipelines (ECDSA, Schnorr)
  char *pub, *A, *B, *R, *u1, *u2, *pub33, *hash;
  uint8_t *pub_ok, *pub_inf, *pre_ok, *A_inf, *B_inf, *R_inf;
};
static int sig_ws(ncg_ctx* ctx, size_t n, hipStream_t st, SigWs* w) {
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t pt_b = al(n * 64), sc_b = al(n * 32), fl_b = al(n), pk_b = al(n * 33);
  const size_t need = 4 * pt_b + 3 * sc_b + 6 * fl_b + pk_b;
  if (ctx->ecdsa_ws_bytes < need) {
    NCG_HIP(ctx, hipStreamSynchronize(st));
    if (ctx->ecdsa_ws) (void)hipFree(ctx->ecdsa_ws);
    ctx->ecdsa_ws = nullptr;
    ctx->ecdsa_ws_bytes = 0;
    hipError_t e = hipMalloc(&ctx->ecdsa_ws, need + (need >> 2));
    if (e != hipSuccess) return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
    ctx->ecdsa_ws_bytes = need + (need >> 2);
  }
  char* p = (char*)ctx->ecdsa_ws;
  w->pub = p;    p += pt_b;
  w->A = p;      p += pt_b;
  w->B = p;      p += pt_b;
  w->R = p;      p += pt_b;
  w->u1 = p;     p += sc_b;
  w->u2 = p;     p += sc_b;
  w->pub33 = p;  p += pk_b;
  w->hash = p;   p += sc_b;
  w->pub_ok = (uint8_t*)p;   p += fl_b;
  w->pub_inf = (uint8_t*)p;  p += fl_b;
  w->pre_ok = (uint8_t*)p;   p += fl_b;
  w->A_inf = (uint8_t*)p;    p += fl_b;
  w->B_inf = (uint8_t*)p;    p += fl_b;
  w->R_inf = (uint8_t*)p;
  return NCG_OK;
}
// R = u1 G + u2 P for decoded keys: fixed-base table, variable-base ladder, one pairwise add
static int sig_mul_add(ncg_ctx* ctx, int curve, size_t n, const SigWs& w, hipStream_t st) {
  int rc = ncg_mul_base_batch_dev(ctx, curve, n, w.u1, w.A, w.A_inf, st);
  if (rc) return rc;
  rc = ncg_mul_var_batch_dev(ctx, curve, n, w.pub, w.u2, w.B, w.B_inf, st);  // rejected keys decode to (0,0) = O
  if (rc) return rc;
  return ncg_add_pairs_batch_dev(ctx, curve, n, w.A, w.B, 0, w.R, w.R_inf, st);
}

int ncg_ecdsa_verify_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* sig64_dev, const void* hash32_dev,
                               const void* pub33_dev, int flags, uint8_t* out_ok_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (curve != NCG_SECP256K1) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: ecdsa_verify: secp256k1 only");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig64_dev || !hash32_dev || !pub33_dev || !out_ok_dev) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ecdsa_verify: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SigWs w;
  int rc = sig_ws(ctx, n, st, &w);
  if (rc) return rc;
  if (flags & NCG_ECDSA_PUB_UNCOMPRESSED) {  // 65-byte keys: range + curve-equation check, no square root
    NCG_HIP(ctx, ncg::secp_load_uncompressed((const uint8_t*)pub33_dev, (uint32_t*)w.pub, w.pub_ok, w.pub_inf, (int)n, st));
  } else {
    rc = ncg_decode_points_batch_dev(ctx, curve, n, pub33_dev, 0, w.pub, w.pub_ok, w.pub_inf, st);
    if (rc) return rc;
  }
  NCG_HIP(ctx, ncg::ecdsa_prepare((const uint8_t*)sig64_dev, (const uint8_t*)hash32_dev, (int)n, (flags & NCG_ECDSA_LOW_S) != 0,
                                  (uint32_t*)w.u1, (uint32_t*)w.u2, w.pre_ok, st));
  rc = sig_mul_add(ctx, curve, n, w, st);
  if (rc) return rc;
  NCG_HIP(ctx, ncg::ecdsa_finish((const uint8_t*)sig64_dev, (const uint32_t*)w.R, w.R_inf, w.pre_ok, w.pub_ok, w.pub_inf, (int)n,
                                 out_ok_dev, st));
  return NCG_OK;
}

// Q[i] = recoverPublicKey(sig65[i], hash[i]) as an affine wire point, out_ok[i] = 0 where the reference throws
int ncg_ecdsa_recover_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* sig65_dev, const void* hash32_dev,
                                void* out_affine_dev, uint8_t* out_ok_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (curve != NCG_SECP256K1) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: ecdsa_recover: secp256k1 only");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig65_dev || !hash32_dev || !out_affine_dev || !out_ok_dev) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ecdsa_recover: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SigWs w;
  int rc = sig_ws(ctx, n, st, &w);
  if (rc) return rc;
  NCG_HIP(ctx, ncg::ecdsa_recover_prepare((const uint8_t*)sig65_dev, (const uint8_t*)hash32_dev, (int)n, (uint32_t*)w.u1, (uint32_t*)w.u2,
                                          (uint8_t*)w.pub33, w.pre_ok, st));
  rc = ncg_decode_points_batch_dev(ctx, curve, n, w.pub33, 0, w.pub, w.pub_ok, w.pub_inf, st);  // R from (x, parity)
  if (rc) return rc;
  rc = ncg_mul_base_batch_dev(ctx, curve, n, w.u1, w.A, w.A_inf, st);
  if (rc) return rc;
  rc = ncg_mul_var_batch_dev(ctx, curve, n, w.pub, w.u2, w.B, w.B_inf, st);
  if (rc) return rc;
  rc = ncg_add_pairs_batch_dev(ctx, curve, n, w.A, w.B, 0, out_affine_dev, w.R_inf, st);
  if (rc) return rc;
  NCG_HIP(ctx, ncg::ecdsa_recover_finish((uint32_t*)out_affine_dev, w.R_inf, w.pre_ok, w.pub_ok, (int)n, out_ok_dev, st));
  return NCG_OK;
}

int ncg_ecdsa_recover_batch(ncg_ctx* ctx, int curve, size_t n, const void* sig65, const void* hash32, void* out_affine,
                            uint8_t* out_ok) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (curve != NCG_SECP256K1) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: ecdsa_recover: secp256k1 only");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig65 || !hash32 || !out_affine || !out_ok) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ecdsa_recover: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  int rc = ensure_scratch(ctx, al(n * 65) + al(n * 32) + al(n * 64) + al(n) + 1024);
  if (rc) return rc;
  char* d_sig = (char*)ctx->scratch;
  char* d_hash = d_sig + al(n * 65);
  char* d_out = d_hash + al(n * 32);
  char* d_ok = d_out + al(n * 64);
  NCG_HIP(ctx, pins.h2d(d_sig, sig65, n * 65));
  NCG_HIP(ctx, pins.h2d(d_hash, hash32, n * 32));
  rc = ncg_ecdsa_recover_batch_dev(ctx, curve, n, d_sig, d_hash, d_out, (uint8_t*)d_ok, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_affine, d_out, n * 64));
  NCG_HIP(ctx, pins.d2h(out_ok, d_ok, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

int ncg_schnorr_verify_batch_dev(ncg_ctx* ctx, size_t n, const void* sig64_dev, const void* e32_dev, const void* pkx32_dev,
                                 uint8_t* out_ok_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig64_dev || !e32_dev || !pkx32_dev || !out_ok_dev) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: schnorr_verify: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SigWs w;
  int rc = sig_ws(ctx, n, st, &w);
  if (rc) return rc;
  NCG_HIP(ctx, ncg::schnorr_prepare((const uint8_t*)sig64_dev, (const uint8_t*)e32_dev, (const uint8_t*)pkx32_dev, (int)n,
                                    (uint32_t*)w.u1, (uint32_t*)w.u2, (uint8_t*)w.pub33, w.pre_ok, st));
  rc = ncg_decode_points_batch_dev(ctx, NCG_SECP256K1, n, w.pub33, 0, w.pub, w.pub_ok, w.pub_inf, st);  // lift_x: the even root
  if (rc) return rc;
  rc = sig_mul_add(ctx, NCG_SECP256K1, n, w, st);
  if (rc) return rc;
  NCG_HIP(ctx, ncg::schnorr_finish((const uint8_t*)sig64_dev, (const uint32_t*)w.R, w.R_inf, w.pre_ok, w.pub_ok, w.pub_inf, (int)n,
                                   out_ok_dev, st));
  return NCG_OK;
}

// ---- the same from messages: SHA-256 on the device (csrc/sha256.hpp) - the prehash of ecdsa.verify and the BIP-340
// tagged challenge - then the entry points above
int ncg_ecdsa_verify_batch_msgs_dev(ncg_ctx* ctx, int curve, size_t n, const void* sig64_dev, const void* msgs_dev,
                                    const uint64_t* msg_off_dev, const void* pub_dev, int flags, uint8_t* out_ok_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (curve != NCG_SECP256K1) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: ecdsa_verify: secp256k1 only");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig64_dev || !msg_off_dev || !pub_dev || !out_ok_dev) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ecdsa_verify_msgs: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SigWs w;
  int rc = sig_ws(ctx, n, st, &w);
  if (rc) return rc;
  NCG_HIP(ctx, ncg::sha256_msgs((const uint8_t*)msgs_dev, msg_off_dev, nullptr, nullptr, 0, (int)n, (uint8_t*)w.hash, st));
  return ncg_ecdsa_verify_batch_dev(ctx, curve, n, sig64_dev, w.hash, pub_dev, flags, out_ok_dev, st);
}

int ncg_schnorr_verify_batch_msgs_dev(ncg_ctx* ctx, size_t n, const void* sig64_dev, const void* msgs_dev,
                                      const uint64_t* msg_off_dev, const void* pkx32_dev, uint8_t* out_ok_dev, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig64_dev || !msg_off_dev || !pkx32_dev || !out_ok_dev) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: schnorr_verify_msgs: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SigWs w;
  int rc = sig_ws(ctx, n, st, &w);
  if (rc) return rc;
  NCG_HIP(ctx, ncg::sha256_msgs((const uint8_t*)msgs_dev, msg_off_dev, (const uint8_t*)sig64_dev, (const uint8_t*)pkx32_dev, 1, (int)n,
                                (uint8_t*)w.hash, st));
  return ncg_schnorr_verify_batch_dev(ctx, n, sig64_dev, w.hash, pkx32_dev, out_ok_dev, st);
}

// host-pointer variants: mode 0 = ECDSA (keys: kb bytes per row), mode 1 = Schnorr (32-byte x-only keys)
static int sig_verify_msgs_host(ncg_ctx* ctx, int mode, size_t n, const void* sig64, const void* msgs, const uint64_t* msg_off,
                                const void* keys, size_t kb, int flags, uint8_t* out_ok) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig64 || !msg_off || !keys || !out_ok) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: verify_msgs: NULL buffer");
  for (size_t i = 0; i < n; i++)
    if (msg_off[i + 1] < msg_off[i]) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: verify_msgs: offsets must not decrease (index %zu)", i);
  const size_t mbytes = (size_t)(msg_off[n] - msg_off[0]);
  if (mbytes && !msgs) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: verify_msgs: NULL message buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t off_b = (n + 1) * 8;
  int rc = ensure_scratch(ctx, al(n * 64) + al(n * kb) + al(off_b) + al(mbytes + 8) + al(n) + 1024);
  if (rc) return rc;
  char* d_sig = (char*)ctx->scratch;
  char* d_key = d_sig + al(n * 64);
  char* d_off = d_key + al(n * kb);
  char* d_msg = d_off + al(off_b);
  char* d_ok = d_msg + al(mbytes + 8);
  NCG_HIP(ctx, pins.h2d(d_sig, sig64, n * 64));
  NCG_HIP(ctx, pins.h2d(d_key, keys, n * kb));
  std::vector<uint64_t> rel(n + 1);
  for (size_t i = 0; i <= n; i++) rel[i] = msg_off[i] - msg_off[0];
  NCG_HIP(ctx, hipMemcpyAsync(d_off, rel.data(), off_b, hipMemcpyHostToDevice, ctx->stream));
  if (mbytes) NCG_HIP(ctx, pins.h2d(d_msg, (const char*)msgs + msg_off[0], mbytes));
  if (mode == 0)
    rc = ncg_ecdsa_verify_batch_msgs_dev(ctx, NCG_SECP256K1, n, d_sig, d_msg, (const uint64_t*)d_off, d_key, flags, (uint8_t*)d_ok, ctx->stream);
  else
    rc = ncg_schnorr_verify_batch_msgs_dev(ctx, n, d_sig, d_msg, (const uint64_t*)d_off, d_key, (uint8_t*)d_ok, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_ok, d_ok, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));  // also keeps `rel` alive until its copy is done
  return NCG_OK;
}
int ncg_ecdsa_verify_batch_msgs(ncg_ctx* ctx, int curve, size_t n, const void* sig64, const void* msgs, const uint64_t* msg_off,
                                const void* pub, int flags, uint8_t* out_ok) {
  if (ctx && curve != NCG_SECP256K1) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: ecdsa_verify: secp256k1 only");
  return sig_verify_msgs_host(ctx, 0, n, sig64, msgs, msg_off, pub, (flags & NCG_ECDSA_PUB_UNCOMPRESSED) ? 65 : 33, flags, out_ok);
}
int ncg_schnorr_verify_batch_msgs(ncg_ctx* ctx, size_t n, const void* sig64, const void* msgs, const uint64_t* msg_off,
                                  const void* pkx32, uint8_t* out_ok) {
  return sig_verify_msgs_host(ctx, 1, n, sig64, msgs, msg_off, pkx32, 32, 0, out_ok);
}

int ncg_schnorr_verify_batch(ncg_ctx* ctx, size_t n, const void* sig64, const void* e32, const void* pkx32, uint8_t* out_ok) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig64 || !e32 || !pkx32 || !out_ok) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: schnorr_verify: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  int rc = ensure_scratch(ctx, al(n * 64) + 2 * al(n * 32) + al(n) + 1024);
  if (rc) return rc;
  char* d_sig = (char*)ctx->scratch;
  char* d_e = d_sig + al(n * 64);
  char* d_pk = d_e + al(n * 32);
  char* d_ok = d_pk + al(n * 32);
  NCG_HIP(ctx, pins.h2d(d_sig, sig64, n * 64));
  NCG_HIP(ctx, pins.h2d(d_e, e32, n * 32));
  NCG_HIP(ctx, pins.h2d(d_pk, pkx32, n * 32));
  rc = ncg_schnorr_verify_batch_dev(ctx, n, d_sig, d_e, d_pk, (uint8_t*)d_ok, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_ok, d_ok, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

int ncg_ecdsa_verify_batch(ncg_ctx* ctx, int curve, size_t n, const void* sig64, const void* hash32, const void* pub33, int flags,
                           uint8_t* out_ok) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (curve != NCG_SECP256K1) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: ecdsa_verify: secp256k1 only");
  if (n == 0) return NCG_OK;
  if (n > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: batch too large");
  if (!sig64 || !hash32 || !pub33 || !out_ok) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ecdsa_verify: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  PinSet pins(ctx);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t kb = (flags & NCG_ECDSA_PUB_UNCOMPRESSED) ? 65 : 33;  // bytes per key row
  int rc = ensure_scratch(ctx, al(n * 64) + al(n * 32) + al(n * kb) + al(n) + 1024);
  if (rc) return rc;
  char* d_sig = (char*)ctx->scratch;
  char* d_hash = d_sig + al(n * 64);
  char* d_pub = d_hash + al(n * 32);
  char* d_ok = d_pub + al(n * kb);
  NCG_HIP(ctx, pins.h2d(d_sig, sig64, n * 64));
  NCG_HIP(ctx, pins.h2d(d_hash, hash32, n * 32));
  NCG_HIP(ctx, pins.h2d(d_pub, pub33, n * kb));
  rc = ncg_ecdsa_verify_batch_dev(ctx, curve, n, d_sig, d_hash, d_pub, flags, (uint8_t*)d_ok, ctx->stream);
  if (rc) return rc;
  NCG_HIP(ctx, pins.d2h(out_ok, d_ok, n));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

int ncg_field_check(ncg_ctx* ctx, int field, int op, int variant, size_t n, const void* a, const void* b, void* out) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (field < 0 || field > 4) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: field_check: unknown field %d", field);
  if (n == 0) return NCG_OK;
  if (n > (1u << 24) || !a || !b || !out) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: field_check: bad arguments");
  // words per item: fe9 9 in / 8 out; Fe29 from wire 12 / 12; Fe29 raw limbs [a, c] 28 / 12; lane-paired Fp2 raw [a, c] 56 / 24
  const size_t in_w = field == 4 ? 56 : field == 3 ? 28 : field == 2 ? 12 : 9, out_w = field == 4 ? 24 : field >= 2 ? 12 : 8;
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  const size_t in_b = (n * in_w * 4 + 255) & ~(size_t)255, out_b = n * out_w * 4;
  int rc = ensure_scratch(ctx, 2 * in_b + out_b + 1024);
  if (rc) return rc;
  char* d_a = (char*)ctx->scratch;
  char* d_b = d_a + in_b;
  char* d_o = d_b + in_b;
  NCG_HIP(ctx, hipMemcpyAsync(d_a, a, n * in_w * 4, hipMemcpyHostToDevice, ctx->stream));
  NCG_HIP(ctx, hipMemcpyAsync(d_b, b, n * in_w * 4, hipMemcpyHostToDevice, ctx->stream));
  NCG_HIP(ctx, hipMemsetAsync(d_o, 0, out_b, ctx->stream));
  NCG_HIP(ctx, ncg::field_check_run(field, op, variant, (const uint32_t*)d_a, (const uint32_t*)d_b, (uint32_t*)d_o, (int)n, ctx->stream));
  NCG_HIP(ctx, hipMemcpyAsync(out, d_o, out_b, hipMemcpyDeviceToHost, ctx->stream));
  NCG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NCG_OK;
}

int ncg_ubench(ncg_ctx* ctx, int kind, int blocks, int threads, int iters, float* out_ms) {
  if (!ctx || !out_ms) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ubench: NULL arg");
  if (blocks <= 0 || threads <= 0 || threads > 256 || iters <= 0)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: ubench: bad geometry");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->ub_in) {
    NCG_HIP(ctx, hipMalloc((void**)&ctx->ub_in, 1024 * 4));
    uint32_t h[1024];
    uint32_t x = 0x9e3779b9u;
    for (int i = 0; i < 1024; i++) {
      x ^= x << 13;
      x ^= x >> 17;
      x ^= x << 5;
      h[i] = x;
    }
    NCG_HIP(ctx, hipMemcpy(ctx->ub_in, h, sizeof h, hipMemcpyHostToDevice));
  }
  size_t words = (size_t)blocks * threads;
  if (ctx->ub_out_words < words) {
    if (ctx->ub_out) (void)hipFree(ctx->ub_out);
    ctx->ub_out = nullptr;
    NCG_HIP(ctx, hipMalloc((void**)&ctx->ub_out, words * 4));
    ctx->ub_out_words = words;
  }
  NCG_HIP(ctx, ncg::ubench_run(kind, blocks, threads, iters, ctx->ub_out, ctx->ub_in, ctx->stream, out_ms));
  return NCG_OK;
}

}  // extern "C"
#pragma GCC visibility pop
