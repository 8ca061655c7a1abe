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

int ensure_comm_buf(ncg_ctx* ctx, size_t bytes) {
  if (ctx->comm_buf_bytes >= bytes) return NCG_OK;
  if (ctx->comm_buf) (void)hipFree(ctx->comm_buf);
  ctx->comm_buf = nullptr;
  ctx->comm_buf_bytes = 0;
  hipError_t e = hipMalloc(&ctx->comm_buf, bytes);
  if (e != hipSuccess) return set_err(ctx, NCG_ERR_NOMEM, "noble-gpu: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  ctx->comm_buf_bytes = bytes;
  return NCG_OK;
}

// Window plan shared by all shards of one MSM: every rank must cut the same windows, so the width is
// chosen for the LARGEST shard (the caller passes it) - which also re-tunes c for the shard size:
// the bucket fold costs ~2^c per window whatever the shard holds.  Slot format: msm_shard.hpp.
size_t comm_buf_need(size_t stride, size_t fin_words, int nparts);

// this rank's contribution = header + grouped window sums, written at comm_buf + rank * stride
int local_phase(ncg_ctx* ctx, int curve, size_t n_local, size_t n_plan, const void* d_pts, const void* d_sc, int slot,
                int nslots, ncg::MsmPlan* pl_out, size_t* stride_out, hipStream_t st, const uint32_t** d_bad_out = nullptr) {
  if (d_bad_out) *d_bad_out = nullptr;
  ncg::MsmPlan pl;
  int rc = ncg_msm_plan_ws(ctx, curve, n_plan, 0, &pl);
  if (rc) return rc;
  const int c = pl.c;
  if (n_local != n_plan && n_local > 0) {
    rc = ncg_msm_plan_ws(ctx, curve, n_local, c, &pl);  // same windows, this shard's size
    if (rc) return rc;
  }
  const size_t fin_bytes = ncg::msm_fin_words(curve, pl) * 4;
  const size_t stride = ncg::msm_shard_slot_bytes(curve);  // the same on every rank, whatever its plan
  rc = ensure_comm_buf(ctx, comm_buf_need(stride, ncg::msm_shard_max_fin_words(curve), nslots));
  if (rc) return rc;
  char* mine = (char*)ctx->comm_buf + stride * (size_t)slot;
  FinHeader h{(uint32_t)pl.c, (uint32_t)pl.nwin, (uint32_t)(fin_bytes / 4), (uint32_t)curve};
  NCG_HIP(ctx, hipMemcpyAsync(mine, &h, sizeof h, hipMemcpyHostToDevice, st));
  if (n_local == 0) {  // an empty shard contributes identities (all-zero accumulators decode as such)
    NCG_HIP(ctx, hipMemsetAsync(mine + sizeof h, 0, fin_bytes, st));
  } else {
    const uint32_t *d_fin = nullptr, *d_bad = nullptr;
    NCG_HIP(ctx, ncg::msm_device_phase(curve, pl, (const uint32_t*)d_pts, (const uint32_t*)d_sc, ctx->msm_ws, &d_fin, st, &d_bad, &ctx->msm_side));
    if (d_bad_out) *d_bad_out = d_bad;
    NCG_HIP(ctx, hipMemcpyAsync(mine + sizeof h, d_fin, fin_bytes, hipMemcpyDeviceToDevice, st));
  }
  *pl_out = pl;
  *stride_out = stride;
  return NCG_OK;
}

// after the gather: check the headers, add the nparts arrays, finish.  comm_buf layout: nparts slots of
// `stride` bytes, then one more slot for the sum.
int combine_and_finish(ncg_ctx* ctx, int curve, const ncg::MsmPlan& pl, size_t stride, int nparts, void* out_affine,
                       uint8_t* out_is_inf, hipStream_t st) {
  const size_t fin_words = ncg::msm_fin_words(curve, pl);
  char* base = (char*)ctx->comm_buf;
  // the headers travel to the host (checked after the stream is drained); the payloads are compacted into a
  // contiguous [nparts][fin_words] scratch behind the slots, which the adding kernel reduces in place
  std::vector<FinHeader> hs(nparts);
  for (int r = 0; r < nparts; r++)
    NCG_HIP_DRAIN(ctx, st, hipMemcpyAsync(&hs[r], base + stride * (size_t)r, sizeof(FinHeader), hipMemcpyDeviceToHost, st));
  // pack payloads: [nparts][fin_words] behind the slots (sized by comm_buf_need)
  uint32_t* packed = (uint32_t*)(base + stride * (size_t)nparts);
  for (int r = 0; r < nparts; r++)
    NCG_HIP_DRAIN(ctx, st, hipMemcpyAsync(packed + (size_t)r * fin_words, base + stride * (size_t)r + sizeof(FinHeader), fin_words * 4,
                                          hipMemcpyDeviceToDevice, st));
  uint32_t* sum = packed + (size_t)nparts * fin_words;
  const size_t npoints = fin_words / ncg::msm_acc_words(curve);
  NCG_HIP_DRAIN(ctx, st, ncg::msm_sum_partials(curve, packed, nparts, npoints, sum, st));
  uint8_t inf_local = 0;
  NCG_HIP_DRAIN(ctx, st, ncg::msm_finish(curve, pl, sum, (uint32_t*)out_affine, &inf_local, st));  // synchronises st
  char msg[320];
  if (ncg::msm_shard_check(hs.data(), nparts, curve, pl, fin_words, msg, sizeof msg) >= 0) return set_err(ctx, NCG_ERR_INVALID_ARG, "%s", msg);
  if (out_is_inf) *out_is_inf = inf_local;
  return NCG_OK;
}

size_t comm_buf_need(size_t stride, size_t fin_words, int nparts) {
  return stride * (size_t)nparts + ((size_t)nparts + 1) * fin_words * 4 + 256;
}

int identity_out(int curve, void* out_affine, uint8_t* out_is_inf) {
  const int pb = ncg_point_bytes(curve);
  memset(out_affine, 0, pb);
  if (curve == NCG_ED25519) ((uint8_t*)out_affine)[32] = 1;
  if (out_is_inf) *out_is_inf = 1;
  return NCG_OK;
}

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

int ncg_msm_sharded_dev(ncg_ctx* ctx, int curve, size_t n_local, size_t n_max, const void* points_affine_dev,
                        const void* scalars_dev, void* out_affine, uint8_t* out_is_inf, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (ncg_point_bytes(curve) == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: msm_sharded: unsupported curve %d", curve);
  if (!out_affine) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_sharded: NULL output");
  if (n_max == 0) n_max = n_local;
  if (n_local > n_max || n_max > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_sharded: n_local > n_max");
  if (n_local && (!points_affine_dev || !scalars_dev)) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_sharded: NULL buffer");
  const int G = ctx->comm ? ctx->comm_size : 1;
  if (n_max == 0) return identity_out(curve, out_affine, out_is_inf);  // every shard empty (curve.ts:878)
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  ncg::MsmPlan pl;
  size_t stride = 0;
  {  // size the gather buffer before anything is enqueued
    int rc = ensure_comm_buf(ctx, comm_buf_need(ncg::msm_shard_slot_bytes(curve), ncg::msm_shard_max_fin_words(curve), G));
    if (rc) return rc;
  }
  const uint32_t* d_bad = nullptr;
  int rc = local_phase(ctx, curve, n_local, n_max, points_affine_dev, scalars_dev, ctx->comm ? ctx->comm_rank : 0, G, &pl, &stride, st,
                       &d_bad);
  if (rc) return rc;
  if (ctx->comm && G > 1) {
    const Rccl* r = rccl();
    if (!r) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: librccl.so not found");
    char* base = (char*)ctx->comm_buf;
    // in place: rank r's slot already sits at offset r * stride of the receive buffer
    NCG_NCCL(ctx, r, r->AllGather(base + stride * (size_t)ctx->comm_rank, base, stride, ncclUint8, (ncclComm_t)ctx->comm, st));
  }
  uint32_t bad = 0xFFFFFFFFu;  // this rank's scalar-range verdict (validateMSMScalars, curve.ts:398-404); read in stream order
  if (d_bad) NCG_HIP(ctx, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
  rc = combine_and_finish(ctx, curve, pl, stride, G, out_affine, out_is_inf, st);  // drains st on every path
  if (rc) return rc;
  if (bad != 0xFFFFFFFFu)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_sharded: invalid scalar at index %u of this rank's shard (not below the group order)", bad);
  return NCG_OK;
}

// ---- host-staged exchange: the same sharded MSM for transports other than RCCL (gloo, MPI, sockets) --------
// ncg_msm_shard_local_dev runs this rank's phase and returns its slot (header + grouped window sums,
// ncg_msm_shard_slot_bytes(curve) bytes) in host memory; the caller moves the slots of all ranks by whatever
// means it has and hands them to ncg_msm_shard_combine on any rank, which uploads them and runs the same
// header check, adding kernel and finish that ncg_msm_sharded_dev runs after its all-gather.
size_t ncg_msm_shard_slot_bytes(int curve) { return ncg_point_bytes(curve) ? ncg::msm_shard_slot_bytes(curve) : 0; }

int ncg_msm_shard_local_dev(ncg_ctx* ctx, int curve, size_t n_local, size_t n_max, const void* points_affine_dev,
                            const void* scalars_dev, void* slot_out, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (ncg_point_bytes(curve) == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: msm_shard_local: unsupported curve %d", curve);
  if (!slot_out) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_local: NULL output");
  if (n_max == 0) n_max = n_local;
  if (n_local > n_max || n_max > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_local: n_local > n_max");
  if (n_local && (!points_affine_dev || !scalars_dev)) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_local: NULL buffer");
  const size_t slot_bytes = ncg::msm_shard_slot_bytes(curve);
  memset(slot_out, 0, slot_bytes);
  if (n_max == 0) return NCG_OK;  // every shard empty: an all-zero slot (combine returns the identity for n_max = 0)
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  ncg::MsmPlan pl;
  size_t stride = 0;
  const uint32_t* d_bad = nullptr;
  int rc = local_phase(ctx, curve, n_local, n_max, points_affine_dev, scalars_dev, 0, 1, &pl, &stride, st, &d_bad);
  if (rc) return rc;
  uint32_t bad = 0xFFFFFFFFu;
  if (d_bad) NCG_HIP_DRAIN(ctx, st, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
  NCG_HIP_DRAIN(ctx, st, hipMemcpyAsync(slot_out, ctx->comm_buf, sizeof(FinHeader) + ncg::msm_fin_words(curve, pl) * 4, hipMemcpyDeviceToHost, st));
  NCG_HIP(ctx, hipStreamSynchronize(st));
  if (bad != 0xFFFFFFFFu)
    return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_sharded: invalid scalar at index %u of this rank's shard (not below the group order)", bad);
  return NCG_OK;
}

int ncg_msm_shard_combine(ncg_ctx* ctx, int curve, size_t n_max, int nparts, const void* slots, void* out_affine, uint8_t* out_is_inf,
                          void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  if (ncg_point_bytes(curve) == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: msm_shard_combine: unsupported curve %d", curve);
  if (!out_affine || !slots || nparts < 1 || nparts > 4096) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_combine: bad arguments");
  if (n_max > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_shard_combine: too many points");
  if (n_max == 0) return identity_out(curve, out_affine, out_is_inf);  // every shard empty (curve.ts:878)
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  ncg::MsmPlan pl;
  if (ncg::msm_make_plan(curve, (int)n_max, 0, &pl) != 0) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm: cannot plan windows");
  const size_t stride = ncg::msm_shard_slot_bytes(curve);
  int rc = ensure_comm_buf(ctx, comm_buf_need(stride, ncg::msm_shard_max_fin_words(curve), nparts));
  if (rc) return rc;
  NCG_HIP(ctx, hipMemcpyAsync(ctx->comm_buf, slots, stride * (size_t)nparts, hipMemcpyHostToDevice, st));
  return combine_and_finish(ctx, curve, pl, stride, nparts, out_affine, out_is_inf, st);
}

// The sharded pipeline on ONE GPU (self-check and A/B of the shard-size plan): the point set is cut into
// `parts` slices, each runs the per-shard phase in turn, and the slices' window sums go through the same
// combine kernel and finish that G GPUs use - everything of ncg_msm_sharded_dev except the all-gather.
int ncg_msm_split_dev(ncg_ctx* ctx, int curve, size_t n, int parts, const void* points_affine_dev, const void* scalars_dev,
                      void* out_affine, uint8_t* out_is_inf, void* stream) {
  if (!ctx) return set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: ctx is NULL");
  const int pb = ncg_point_bytes(curve);
  if (pb == 0) return set_err(ctx, NCG_ERR_UNSUPPORTED, "noble-gpu: msm_split: unsupported curve %d", curve);
  if (!out_affine || parts < 1 || parts > 64) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_split: bad arguments");
  if (n == 0) return identity_out(curve, out_affine, out_is_inf);
  if (!points_affine_dev || !scalars_dev) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_split: NULL buffer");
  NCG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const size_t per = (n + parts - 1) / parts;
  if (per > 0x7fffffffu) return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_split: too many points");
  ncg::MsmPlan pl;
  int rc = ensure_comm_buf(ctx, comm_buf_need(ncg::msm_shard_slot_bytes(curve), ncg::msm_shard_max_fin_words(curve), parts));
  if (rc) return rc;
  size_t stride = 0;
  for (int g = 0; g < parts; g++) {
    const size_t lo = std::min(n, per * (size_t)g), cnt = std::min(n, lo + per) - lo;
    ncg::MsmPlan plg;
    const uint32_t* d_bad = nullptr;
    rc = local_phase(ctx, curve, cnt, per, (const char*)points_affine_dev + lo * (size_t)pb, (const char*)scalars_dev + lo * 32, g,
                     parts, &plg, &stride, st, &d_bad);
    if (rc) return rc;
    if (g == 0) pl = plg;
    if (d_bad) {  // the workspace (and its flag) is reused by the next slice: read it now
      uint32_t bad = 0xFFFFFFFFu;
      NCG_HIP(ctx, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
      NCG_HIP(ctx, hipStreamSynchronize(st));
      if (bad != 0xFFFFFFFFu)
        return set_err(ctx, NCG_ERR_INVALID_ARG, "noble-gpu: msm_split: invalid scalar at index %zu (not below the group order)", lo + bad);
    }
  }
  return combine_and_finish(ctx, curve, pl, stride, parts, out_affine, out_is_inf, st);
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
  ncg::MsmPlan pl;
  size_t stride = 0;
  const Rccl* r = G > 1 ? rccl() : nullptr;
  if (G > 1 && !r) return fail(set_err(nullptr, NCG_ERR_UNSUPPORTED, "noble-gpu: librccl.so not found"));
  std::vector<const uint32_t*> d_bads(G, nullptr);
  std::vector<uint32_t> bads(G, 0xFFFFFFFFu);
  // 1. every device: upload its slice and run the device phase (all asynchronous, one host thread)
  for (int g = 0; g < G; g++) {
    ncg_ctx* ctx = m->ctx[g];
    const size_t lo = std::min(n, per * (size_t)g), cnt = std::min(n, lo + per) - lo;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(set_err(ctx, NCG_ERR_HIP, "noble-gpu: hipSetDevice failed"));
    int rc = ensure_comm_buf(ctx, comm_buf_need(ncg::msm_shard_slot_bytes(curve), ncg::msm_shard_max_fin_words(curve), G));
    if (rc) return fail(rc);
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
    ncg::MsmPlan plg;
    rc = local_phase(ctx, curve, cnt, per, d_pts, d_sc, g, G, &plg, &stride, ctx->stream, &d_bads[g]);
    if (rc) return fail(rc);
    if (g == 0) pl = plg;
  }
  // 2. one grouped all-gather over the device set
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
  // 3. device 0 adds the partial arrays and finishes; the others only drain their streams
  if (hipSetDevice(m->ctx[0]->device) != hipSuccess) return fail(set_err(m->ctx[0], NCG_ERR_HIP, "noble-gpu: hipSetDevice failed"));
  int rc = combine_and_finish(m->ctx[0], curve, pl, stride, G, out_affine, out_is_inf, m->ctx[0]->stream);
  for (int g = 0; g < G; g++) {
    (void)hipSetDevice(m->ctx[g]->device);
    if (d_bads[g]) (void)hipMemcpyAsync(&bads[g], d_bads[g], 4, hipMemcpyDeviceToHost, m->ctx[g]->stream);
    (void)hipStreamSynchronize(m->ctx[g]->stream);
  }
  (void)hipSetDevice(m->ctx[0]->device);
  if (rc) return fail(rc);
  for (int g = 0; g < G; g++)
    if (bads[g] != 0xFFFFFFFFu)
      return fail(set_err(nullptr, NCG_ERR_INVALID_ARG, "noble-gpu: msm_multi: invalid scalar at index %zu (not below the group order)",
                          per * (size_t)g + bads[g]));
  return NCG_OK;
}

}  // extern "C"
#pragma GCC visibility pop
