// MSM kernels and the single-GPU driver (see msm.hpp for the pipeline).
#include "msm.hpp"
#include "knobs.hpp"
// This file is compiled TWICE (Makefile): as msm.o with every curve except bls12-381 G1, and - through msm_g1.hip, which
// defines NCG_MSM_TU_G1 and includes it - as msm_g1.o with the G1 instantiations alone, because the two want different
// instruction schedulers: LLVM's max-ILP strategy runs the G1 kernels 1.6 % faster (2^20 MSM 3.55 -> 3.48 ms) and the
// lane-paired G2 kernels 0.9 % slower (3.39 -> 3.42 ms), measured on MI355X, and the strategy is a per-file option.  The
// curve-independent kernels (digits, counting sort) are `static`: each unit has its own copy of those few hundred bytes.

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#include "host_api.hpp"
#include "msm_coop.hpp"
#include "msm_finish.hpp"
#include "msm_plan.hpp"

namespace ncg {

// ------------------------------------------------------------------ 1. wire -> Montgomery
template <class C>
__global__ void __launch_bounds__(256) k_points_to_mont(const uint32_t* __restrict__ pts, uint32_t* __restrict__ out,
                                                        int n) {
  using G = MsmGroup<C>;
  int i = (blockIdx.x * blockDim.x + threadIdx.x) >> LaneShift<C>::value;
  if (i >= n) return;
  G::wire_to_storage(pts + (size_t)i * G::WIRE_AFF, out + (size_t)i * G::AFF_WORDS);
}

// fix-up / fold / grouping kernels: G1 sits 29-43 registers above the 2-waves/SIMD line and is 2-3 %
// faster when asked to fit; the lane-paired G2 kernels are not (measured)
template <class C> struct TailMinWaves { static constexpr int value = 1; };
// (one wave per SIMD - 297 registers, no spills - measured again in round 5: the 2^20 G1 MSM 3.33 -> 3.48 ms; two it stays)
#ifndef NCG_TAIL_MINW_G1
#define NCG_TAIL_MINW_G1 2
#endif
template <> struct TailMinWaves<CurveG1> { static constexpr int value = NCG_TAIL_MINW_G1; };

// ------------------------------------------------------------------ 2. signed digits
// digits[w*n + i] = ((k_i + H') >> (c w)) & (2^c - 1)) - 2^(c-1)
// A scalar >= the group order is outside the reference's contract (validateMSMScalars, curve.ts:398-404:
// 'invalid scalar at index i') and outside the window plan: it is reported through *bad_index (smallest
// offending index) and the call fails.
static __global__ void __launch_bounds__(256) k_msm_digits(const uint32_t* __restrict__ scalars, int16_t* __restrict__ digits,
                                                    MsmPlan pl, uint32_t* __restrict__ bad_index) {
  __shared__ uint32_t sh[256 * 11];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t* my = sh + threadIdx.x * 11;
  if (i < pl.n) {
    uint32_t k[8];  // two 16-byte loads per scalar (the word-by-word form issued sixteen strided loads per lane)
    {
      const uint4* kp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
      const uint4 lo = kp[0], hi = kp[1];
      k[0] = lo.x; k[1] = lo.y; k[2] = lo.z; k[3] = lo.w;
      k[4] = hi.x; k[5] = hi.y; k[6] = hi.z; k[7] = hi.w;
    }
    {
      uint32_t bw = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) (void)__builtin_subc(k[j], pl.order[j], bw, &bw);
      if (bw == 0) atomicMin(bad_index, pl.index_base + (uint32_t)i);  // scalar >= order
    }
    uint32_t cy = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) my[j] = __builtin_addc(k[j], pl.hconst[j], cy, &cy);
    my[8] = __builtin_addc(0u, pl.hconst[8], cy, &cy);
    my[9] = pl.hconst[9] + cy;
    my[10] = 0;
    const uint32_t mask = (1u << pl.c) - 1u;
    const int half = 1 << (pl.c - 1);
    const int nwt = pl.nwin_total ? pl.nwin_total : pl.nwin;
    for (int w = 0; w < pl.nwin; w++) {
      int bp = (w + pl.w0) * pl.c;
      int limb = bp >> 5, sft = bp & 31;
      uint64_t two = ((uint64_t)my[limb + 1] << 32) | my[limb];
      uint32_t v = (uint32_t)(two >> sft) & mask;
      int dg = (int)v - half;
      if (pl.top_tb && w + pl.w0 == nwt - 1 && dg > 0)   // short top window: sub-bucket by the point index (MsmPlan::top_tb)
        dg = (int)((((uint32_t)i & pl.top_submask) << pl.top_tb) + (uint32_t)min(dg, 1 << pl.top_tb));
      digits[(size_t)w * pl.n + i] = (int16_t)dg;
    }
  }
}

// Block -> (chunk q, window w) of the two sort kernels.  With pl.xcd_map the grid is 1-D and the window index is
// the fastest-varying one modulo 8, so every block of a window lands on the same XCD (consecutive workgroup ids
// go round-robin over the 8 XCDs): the window's sorted list (4 MB at 2^20 points) is then written through ONE L2.
__device__ __forceinline__ bool msm_sort_block(const MsmPlan& pl, int& q, int& w) {
  if (pl.xcd_map) {
    const int nw8 = (pl.nwin + 7) & ~7;
    const int id = blockIdx.x;
    w = id % nw8;
    q = id / nw8;
    return w < pl.nwin;
  }
  q = blockIdx.x;
  w = blockIdx.y;
  return true;
}

// ------------------------------------------------------------------ 3. counting sort
// counts[(w*Q + q)*nb + b]: number of entries of chunk q in bucket b (bucket value b+1)
static __global__ void __launch_bounds__(1024) k_msm_hist(const int16_t* __restrict__ digits, uint32_t* __restrict__ counts,
                                                   MsmPlan pl) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
  int q, w;
  if (!msm_sort_block(pl, q, w)) return;
  for (int b = threadIdx.x; b < pl.nb; b += blockDim.x) hist[b] = 0;
  __syncthreads();
  const int lo = q * pl.chunk, hi = min(pl.n, lo + pl.chunk);
  const int16_t* dg = digits + (size_t)w * pl.n;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int d = dg[i];
    if (d != 0) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
  }
  __syncthreads();
  uint32_t* dst = counts + ((size_t)w * pl.Q + q) * pl.nb;
  for (int b = threadIdx.x; b < pl.nb; b += blockDim.x) dst[b] = hist[b];
}

// Per (window, bucket): exclusive prefix of the chunk counts (in place) and the bucket size.
// Adjacent lanes handle adjacent buckets, so every access is coalesced.
static __global__ void __launch_bounds__(256) k_msm_bucket_totals(uint32_t* __restrict__ counts,
                                                           uint32_t* __restrict__ bucket_start, MsmPlan pl) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
  if (b >= pl.nb) return;
  uint32_t* cw = counts + (size_t)w * pl.Q * pl.nb + b;
  uint32_t tot = 0;
  // eight loads in flight per lane: the column walk is a chain of memory latencies otherwise (Q = 128..512 round trips -
  // 64 us of a window-sharded part that runs two windows, one wave per SIMD)
  int q = 0;
  for (; q + 8 <= pl.Q; q += 8) {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = cw[(size_t)(q + j) * pl.nb];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      cw[(size_t)(q + j) * pl.nb] = tot;
      tot += v[j];
    }
  }
  for (; q < pl.Q; q++) {
    uint32_t v = cw[(size_t)q * pl.nb];
    cw[(size_t)q * pl.nb] = tot;
    tot += v;
  }
  bucket_start[(size_t)w * (pl.nb + 1) + b] = tot;  // size for now; k_msm_scan turns it into a start
}

// The same for plans with MANY chunks per window (the ranks of a window-sharded MSM: 2 windows x 256 chunks): a workgroup
// takes 32 adjacent buckets (one 128-byte line per row) and cuts the column walk into 8 slices of Q / 8 rows, one per wave
// quarter; slice totals meet in LDS, then every slice rewrites its rows with its base.  The walk is a chain of memory
// latencies - Q / 8 rounds with eight loads in flight in the one-lane form, 64 us of a 0.88 ms share; here 2 x Q / 64 rounds.
static __global__ void __launch_bounds__(256) k_msm_bucket_totals_split(uint32_t* __restrict__ counts,
                                                                 uint32_t* __restrict__ bucket_start, MsmPlan pl) {
  __shared__ uint32_t part[8][32];
  const int bx = threadIdx.x & 31, qs = threadIdx.x >> 5, w = blockIdx.y;
  const int b = blockIdx.x * 32 + bx;
  const bool live = b < pl.nb;
  const int per = (pl.Q + 7) >> 3;
  const int q0 = min(pl.Q, qs * per), q1 = min(pl.Q, q0 + per);
  uint32_t* cw = counts + (size_t)w * pl.Q * pl.nb + (live ? b : 0);
  uint32_t tot = 0;
  if (live) {
    int q = q0;
    for (; q + 8 <= q1; q += 8) {
      uint32_t v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = cw[(size_t)(q + j) * pl.nb];
#pragma unroll
      for (int j = 0; j < 8; j++) tot += v[j];
    }
    for (; q < q1; q++) tot += cw[(size_t)q * pl.nb];
  }
  part[qs][bx] = tot;
  __syncthreads();
  uint32_t base = 0, all = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t p = part[j][bx];
    if (j < qs) base += p;
    all += p;
  }
  if (!live) return;
  int q = q0;
  for (; q + 8 <= q1; q += 8) {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = cw[(size_t)(q + j) * pl.nb];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      cw[(size_t)(q + j) * pl.nb] = base;
      base += v[j];
    }
  }
  for (; q < q1; q++) {
    const uint32_t v = cw[(size_t)q * pl.nb];
    cw[(size_t)q * pl.nb] = base;
    base += v;
  }
  if (qs == 0) bucket_start[(size_t)w * (pl.nb + 1) + b] = all;  // size for now; k_msm_scan turns it into a start
}

// One block per window: exclusive scan of the bucket sizes -> bucket_start[w][0..nb].  The sizes are staged in
// LDS with coalesced loads (row stride padded by one word per 32 so that the per-thread runs do not share a bank),
// every thread scans its run of nb / 1024 values there, a Hillis-Steele scan joins the per-thread totals.
static __global__ void __launch_bounds__(1024) k_msm_scan(uint32_t* __restrict__ bucket_start, MsmPlan pl) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sizes[];   // nb + nb / 32 words
  __shared__ uint32_t part[1024];
  const int w = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  uint32_t* bs = bucket_start + (size_t)w * (pl.nb + 1);
  for (int b = t; b < pl.nb; b += T) sizes[b + (b >> 5)] = bs[b];
  __syncthreads();
  const int per = (pl.nb + T - 1) / T;
  const int b0 = min(pl.nb, t * per), b1 = min(pl.nb, b0 + per);
  uint32_t mine = 0;
  for (int b = b0; b < b1; b++) mine += sizes[b + (b >> 5)];
  part[t] = mine;
  __syncthreads();
  for (int off = 1; off < T; off <<= 1) {  // inclusive Hillis-Steele scan of per-thread totals
    uint32_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = part[t] - mine;
  for (int b = b0; b < b1; b++) {
    const uint32_t size = sizes[b + (b >> 5)];
    sizes[b + (b >> 5)] = run;
    run += size;
  }
  __syncthreads();
  for (int b = t; b < pl.nb; b += T) bs[b] = sizes[b + (b >> 5)];
  if (t == T - 1) bs[pl.nb] = part[T - 1];
}

// Shared-bucket mode: bucket b of EVERY window is the same bucket.  T_b = sum_w size[w][b] (k_msm_shared_totals),
// exclusive scan over b = the list position of the bucket (k_msm_scan on the one-window array shared_start), then
// every window's share of the bucket gets its start: bucket_start[w][b] = S_b + sum_{w' < w} size[w'][b]
// (k_msm_shared_starts).  shared_start[0 .. nb] = S_b are the accumulate view's starts.
static __global__ void __launch_bounds__(256) k_msm_shared_totals(const uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ shared_start,
                                                           MsmPlan pl) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= pl.nb) return;
  uint32_t tot = 0;
  for (int w = 0; w < pl.nwin; w++) tot += bucket_start[(size_t)w * (pl.nb + 1) + b];
  shared_start[b] = tot;
}
static __global__ void __launch_bounds__(256) k_msm_shared_starts(uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ shared_start,
                                                           MsmPlan pl) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= pl.nb) return;
  uint32_t r = shared_start[b];
  for (int w = 0; w < pl.nwin; w++) {
    uint32_t* p = bucket_start + (size_t)w * (pl.nb + 1) + b;
    const uint32_t size = *p;
    *p = r;
    r += size;
  }
}

// sorted[w*n + pos] = point index | sign<<31, grouped by bucket within the window
// Bucket range [b_lo, b_hi) per launch (round 4): the scattered 4-byte stores are what the kernel costs (220 us of a 2^20-point
// G1 MSM against 45 us with the same LDS atomics and stores that stay in L2 - measured, tools/exp_scatter.py), because a
// window's 4 MB list, written at random positions by blocks on every XCD, never sits in an L2.  Launched once per bucket
// range with the window-major block ids of msm_sort_block (a window's blocks share an XCD), the region a pass writes is
// 4 MB / passes per window - L2-resident - and leaves the cache as whole lines.
static __global__ void __launch_bounds__(1024) k_msm_scatter(const int16_t* __restrict__ digits,
                                                      const uint32_t* __restrict__ counts,
                                                      const uint32_t* __restrict__ bucket_start,
                                                      uint32_t* __restrict__ sorted, MsmPlan pl, int b_lo, int b_hi) {
  extern __shared__ __attribute__((aligned(16))) uint32_t offs_raw[];
  int q, w;
  if (!msm_sort_block(pl, q, w)) return;
  const uint32_t* src = counts + ((size_t)w * pl.Q + q) * pl.nb;
  const uint32_t* bs = bucket_start + (size_t)w * (pl.nb + 1);
  uint32_t* offs = offs_raw - b_lo;   // indexed by bucket
  for (int b = b_lo + threadIdx.x; b < b_hi; b += blockDim.x) offs[b] = src[b] + bs[b];
  __syncthreads();
  const int lo = q * pl.chunk, hi = min(pl.n, lo + pl.chunk);
  const int16_t* dg = digits + (size_t)w * pl.n;
  // shared-bucket mode: one list for all windows (the starts already include the other windows' shares), and the
  // entry names the window's shifted copy of the point
  uint32_t* dst = sorted + (pl.shared ? 0 : (size_t)w * pl.n);
  const uint32_t level = pl.shared ? (uint32_t)(w + pl.w0) * (uint32_t)pl.n : 0u;  // the window's shifted copy (global window index)
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int d = dg[i];
    const int bk = (d < 0 ? -d : d) - 1;
    if (d != 0 && bk >= b_lo && bk < b_hi) {
      uint32_t pos = atomicAdd(&offs[bk], 1u);
      dst[pos] = (level + (uint32_t)i) | (d < 0 ? 0x80000000u : 0u);
    }
  }
}

// ------------------------------------------------------------------ 3b. the counting sort in TWO levels (round 5)
// What the one-level scatter costs is its stores: every entry is a 4-byte write at a random position of its window's 4 MB
// list (16 M different 64-byte sectors per 2^20-point MSM), ~0.2 ms where the same kernel with stores that stay in a cache
// takes 45 us (tools/exp_scatter.py).  Two levels make every store local:
//   k_sort2_count / _scan     per (chunk, window): entries per COARSE RANGE of buckets (64 ranges) and their prefix: region r
//                             of a window's list holds the buckets of range r (128 KB of counts per MSM where the one-level
//                             sort writes, prefixes and re-reads 64 MB of per-chunk bucket counts).
//   k_sort2_scatter           per (chunk, window): every entry goes to ITS REGION of a temporary list, packed with the low
//                             bits of its bucket: a block writes 64 sequential streams - whole lines leave the cache.
//   k_sort2_fine_count / _place   a region is cut into SORT2_SLICES slices, one workgroup each (regions are NOT
//                             equal: every zero scalar lands in the last bucket of every window, bench.py plants one in 17):
//                             slice histograms of the range's buckets in LDS, their prefix (= bucket_start: no separate
//                             totals / scan kernels), then the entries into bucket order - scattered stores again, but
//                             inside one 64 KB span.
// The TOP window of a plan can be short (255 bits = 19 x 13 + 8 for G2 at c = 13: digits 0..232) and its digits are never
// negative: its 64 ranges cover the buckets it can reach only (msm_sort2_ok: top_sh), or four ranges would hold the whole window.
// Same result as hist / bucket_totals / scan / scatter: bucket_start[w][0..nb] and the per-window list grouped by bucket
// (the order inside a bucket differs, as it already does between runs: LDS atomics).  Entries pack index, sign and the low
// bucket bits into 32 bits, so the form is used when they fit (n <= 2^22 at c = 16) and for per-window lists only; other
// plans keep the one-level kernels.
constexpr int SORT2_MAX_RANGES = 256;   // ranges per window: 64, or 128 / 256 where 64 regions would not fit the staged kernel (Sort2::lgr)
constexpr int SORT2_SLICES = 8;
#ifndef NCG_SORT2_UNROLL
#define NCG_SORT2_UNROLL 8
#endif
constexpr int SORT2_UNROLL = NCG_SORT2_UNROLL;   // entries a lane of the coarse kernels holds in flight
struct Sort2 {
  int lgr;                // log2(ranges per window): 6..8
  int sh, idxbits;        // log2(buckets per range), bits of an entry index
  int top_w, top_base, top_sh;   // local index of the plan's top window (-1: not in this plan / full), first bucket its ranges cover, its sh
};
__device__ __forceinline__ void sort2_window(const Sort2& s2, int w, uint32_t& base, int& sh) {
  const bool top = w == s2.top_w;
  base = top ? (uint32_t)s2.top_base : 0u;
  sh = top ? s2.top_sh : s2.sh;
}

// Lanes of a wave that hold the same key serialise on one LDS address (identical scalars, benchmark/bls12-381.ts:64-79: every
// lane, every time; the zero scalars of any input: the last bucket).  When the first key is shared by a quarter of the wave
// the additions are aggregated per distinct key (one atomic per key and wave).  Called by whole waves.
__device__ __forceinline__ bool sort2_heavy(uint32_t key, bool valid, uint64_t& act) {
#ifdef __HIP_DEVICE_COMPILE__
  act = __ballot(valid);
  if (act == 0) return false;
  const uint32_t k0 = (uint32_t)__shfl((int)key, __ffsll((long long)act) - 1);
  return __popcll(__ballot(valid && key == k0)) >= 16;
#else
  return false;
#endif
}
// position from the cursor of `key` for every valid lane
__device__ __forceinline__ uint32_t sort2_claim(uint32_t* ctr, uint32_t key, bool valid) {
#ifdef __HIP_DEVICE_COMPILE__
  uint64_t act;
  uint32_t pos = 0;
  if (sort2_heavy(key, valid, act)) {
    const int lane = (int)(threadIdx.x & 63);
    uint64_t todo = act;
    while (todo) {
      const int f = __ffsll((long long)todo) - 1;
      const uint32_t kf = (uint32_t)__shfl((int)key, f);
      const uint64_t grp = __ballot(valid && key == kf) & todo;
      uint32_t base = 0;
      if (lane == f) base = atomicAdd(&ctr[kf], (uint32_t)__popcll(grp));
      base = (uint32_t)__shfl((int)base, f);
      if ((grp >> lane) & 1) pos = base + (uint32_t)__popcll(grp & ((1ull << lane) - 1ull));
      todo &= ~grp;
    }
  } else if (valid) {
    pos = atomicAdd(&ctr[key], 1u);
  }
  return pos;
#else
  return 0;
#endif
}
// the counting passes: no position comes back (the atomics are not waited for)
__device__ __forceinline__ void sort2_tally(uint32_t* ctr, uint32_t key, bool valid) {
#ifdef __HIP_DEVICE_COMPILE__
  uint64_t act;
  if (sort2_heavy(key, valid, act)) {
    const int lane = (int)(threadIdx.x & 63);
    uint64_t todo = act;
    while (todo) {
      const int f = __ffsll((long long)todo) - 1;
      const uint32_t kf = (uint32_t)__shfl((int)key, f);
      const uint64_t grp = __ballot(valid && key == kf) & todo;
      if (lane == f) atomicAdd(&ctr[kf], (uint32_t)__popcll(grp));
      todo &= ~grp;
    }
  } else if (valid) {
    atomicAdd(&ctr[key], 1u);
  }
#endif
}

// ccount[(w*Q + q)*64 + r]: entries of chunk q whose bucket lies in range r
static __global__ void __launch_bounds__(1024) k_sort2_count(const int16_t* __restrict__ digits, uint32_t* __restrict__ ccount,
                                                      MsmPlan pl, Sort2 s2) {
  __shared__ uint32_t cnt[16][SORT2_MAX_RANGES];   // one copy per wave: the counters are hot
  const int R = 1 << s2.lgr;
  int q, w;
  if (!msm_sort_block(pl, q, w)) return;
  for (int t = threadIdx.x; t < 16 * SORT2_MAX_RANGES; t += blockDim.x) (&cnt[0][0])[t] = 0;
  __syncthreads();
  uint32_t* mine = cnt[threadIdx.x >> 6];
  uint32_t base;
  int sh;
  sort2_window(s2, w, base, sh);
  const int lo = q * pl.chunk, hi = min(pl.n, lo + pl.chunk);
  const int16_t* dg = digits + (size_t)w * pl.n;
  for (int i0 = lo; i0 < hi; i0 += SORT2_UNROLL * (int)blockDim.x) {   // SORT2_UNROLL loads in flight
    int d[SORT2_UNROLL];
#pragma unroll
    for (int k = 0; k < SORT2_UNROLL; k++) {
      const int i = i0 + k * (int)blockDim.x + (int)threadIdx.x;
      d[k] = i < hi ? dg[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < SORT2_UNROLL; k++) {
      const uint32_t bk = (uint32_t)((d[k] < 0 ? -d[k] : d[k]) - 1);
      if (d[k] != 0) atomicAdd(&mine[((bk - base) >> sh) & (R - 1)], 1u);   // (same-key lanes serialise in the LDS unit: 32 rounds per wave, harmless)
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < R) {
    uint32_t t = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) t += cnt[k][threadIdx.x];
    ccount[((size_t)w * pl.Q + q) * R + threadIdx.x] = t;
  }
}

// per window: ccount -> exclusive prefix over the chunks inside each range; region_start[w][0..64] = where range r begins
static __global__ void __launch_bounds__(SORT2_MAX_RANGES) k_sort2_scan(uint32_t* __restrict__ ccount, uint32_t* __restrict__ region_start,
                                                             uint32_t* __restrict__ oversize, MsmPlan pl, Sort2 s2) {
  __shared__ uint32_t tot[SORT2_MAX_RANGES];
  const int w = blockIdx.x, r = threadIdx.x, R = 1 << s2.lgr;   // launched with R lanes
  if (w == 0 && r == 0) oversize[0] = 0;   // the work list of k_sort2_fine_staged
  uint32_t* c = ccount + (size_t)w * pl.Q * R + r;
  uint32_t run = 0;
  int q = 0;
  for (; q + 8 <= pl.Q; q += 8) {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = c[(size_t)(q + j) * R];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c[(size_t)(q + j) * R] = run;
      run += v[j];
    }
  }
  for (; q < pl.Q; q++) {
    const uint32_t v = c[(size_t)q * R];
    c[(size_t)q * R] = run;
    run += v;
  }
  tot[r] = run;
  __syncthreads();
  uint32_t before = 0;
  for (int k = 0; k < r; k++) before += tot[k];
  uint32_t* rs = region_start + (size_t)w * (R + 1);
  rs[r] = before;
  if (r == R - 1) rs[R] = before + run;
}

// tmp[w][region of the entry's range] = index | sign << idxbits | (bucket offset inside the range) << (idxbits + 1)
static __global__ void __launch_bounds__(1024) k_sort2_scatter(const int16_t* __restrict__ digits, const uint32_t* __restrict__ ccount,
                                                        const uint32_t* __restrict__ region_start, uint32_t* __restrict__ tmp,
                                                        MsmPlan pl, Sort2 s2) {
  __shared__ uint32_t cur[SORT2_MAX_RANGES];
  int q, w;
  if (!msm_sort_block(pl, q, w)) return;
  const int R = 1 << s2.lgr;
  if ((int)threadIdx.x < R)
    cur[threadIdx.x] = region_start[(size_t)w * (R + 1) + threadIdx.x] + ccount[((size_t)w * pl.Q + q) * R + threadIdx.x];
  __syncthreads();
  uint32_t base;
  int sh;
  sort2_window(s2, w, base, sh);
  const int lo = q * pl.chunk, hi = min(pl.n, lo + pl.chunk);
  const int16_t* dg = digits + (size_t)w * pl.n;
  uint32_t* dst = tmp + (size_t)w * pl.n;
  const uint32_t lowmask = (1u << sh) - 1u;
  for (int i0 = lo; i0 < hi; i0 += SORT2_UNROLL * (int)blockDim.x) {
    int d[SORT2_UNROLL];
#pragma unroll
    for (int k = 0; k < SORT2_UNROLL; k++) {
      const int i = i0 + k * (int)blockDim.x + (int)threadIdx.x;
      d[k] = i < hi ? dg[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < SORT2_UNROLL; k++) {
      const int i = i0 + k * (int)blockDim.x + (int)threadIdx.x;
      const uint32_t off = (uint32_t)((d[k] < 0 ? -d[k] : d[k]) - 1) - base;
      if (d[k] != 0) dst[atomicAdd(&cur[(off >> sh) & (R - 1)], 1u)] = (uint32_t)i | ((d[k] < 0 ? 1u : 0u) << s2.idxbits) | ((off & lowmask) << (s2.idxbits + 1));
    }
  }
}

// Regions of up to SORT2_STAGE entries (all of them unless the scalars are skewed) are sorted by ONE workgroup with the whole
// region in registers and its output staged in LDS: one coalesced read, one coalesced write, no second pass over memory.
// Larger regions (identical scalars; few distinct values) go through the two slice kernels below, which return at once
// for every region the staged kernel took.
constexpr int SORT2_PER_THREAD = 20;
constexpr int SORT2_STAGE = 1024 * SORT2_PER_THREAD;   // entries: 80 KB of LDS
// bucket starts of range r from the bucket sizes `v` (thread t = bucket t of the range; at most 512 buckets, blockDim >= 512);
// returns this bucket's start.  Also writes what no range covers: the buckets past the top window's last range and the end marker.
__device__ __forceinline__ uint32_t sort2_starts(uint32_t* scan, uint32_t v, uint32_t a, int r, int w, int BL, uint32_t base, int R,
                                                 const uint32_t* region_start, uint32_t* bucket_start, const MsmPlan& pl, bool write = true) {
  const int t = threadIdx.x;
  if (t < 512) scan[t] = v;
  __syncthreads();
  for (int off = 1; off < 512; off <<= 1) {   // inclusive Hillis-Steele scan of the <= 512 bucket sizes
    const uint32_t add = (t < 512 && t >= off) ? scan[t - off] : 0u;
    __syncthreads();
    if (t < 512) scan[t] += add;
    __syncthreads();
  }
  const uint32_t start = t < 512 ? a + scan[t] - v : 0u;
  if (!write) return start;
  uint32_t* bs = bucket_start + (size_t)w * (pl.nb + 1);
  const uint32_t first = base + (uint32_t)r * (uint32_t)BL;   // first bucket of this range
  if (t < BL && first + t < (uint32_t)pl.nb) bs[first + t] = start;
  if (r == 0)
    for (uint32_t b = t; b < base; b += blockDim.x) bs[b] = 0;
  if (r == R - 1) {
    const uint32_t end = region_start[(size_t)w * (R + 1) + R];
    for (uint32_t b = first + (uint32_t)BL + t; b <= (uint32_t)pl.nb; b += blockDim.x) bs[b] = end;
  }
  return start;
}
static __global__ void __launch_bounds__(1024) k_sort2_fine_staged(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ region_start,
                                                            uint32_t* __restrict__ oversize, uint32_t* __restrict__ bucket_start,
                                                            uint32_t* __restrict__ sorted, MsmPlan pl, Sort2 s2) {
  extern __shared__ __attribute__((aligned(16))) uint32_t stage[];   // SORT2_STAGE entries
  __shared__ uint32_t cnt[512];
  __shared__ uint32_t scan[512];
  const int r = blockIdx.x, w = blockIdx.y, t = threadIdx.x, R = 1 << s2.lgr;
  const uint32_t a = region_start[(size_t)w * (R + 1) + r], b = region_start[(size_t)w * (R + 1) + r + 1];
  if (b - a > (uint32_t)SORT2_STAGE) {   // the slice kernels take it
    if (threadIdx.x == 0) oversize[1 + atomicAdd(&oversize[0], 1u)] = ((uint32_t)w << 8) | (uint32_t)r;
    return;
  }
  uint32_t base;
  int sh;
  sort2_window(s2, w, base, sh);
  if (t < 512) cnt[t] = 0;
  __syncthreads();
  const uint32_t* src = tmp + (size_t)w * pl.n;
  uint32_t e[SORT2_PER_THREAD];
#pragma unroll
  for (int k = 0; k < SORT2_PER_THREAD; k++) {
    const uint32_t i = a + (uint32_t)k * 1024u + (uint32_t)t;
    e[k] = i < b ? src[i] : 0u;
  }
#pragma unroll
  for (int k = 0; k < SORT2_PER_THREAD; k++)
    if (a + (uint32_t)k * 1024u + (uint32_t)t < b) atomicAdd(&cnt[e[k] >> (s2.idxbits + 1)], 1u);
  __syncthreads();
  const uint32_t v = t < 512 ? cnt[t] : 0u;
  const uint32_t start = sort2_starts(scan, v, a, r, w, 1 << sh, base, R, region_start, bucket_start, pl);
  __syncthreads();
  if (t < 512) cnt[t] = start - a;   // cursors inside the stage
  __syncthreads();
  const uint32_t idxmask = (1u << s2.idxbits) - 1u;
#pragma unroll
  for (int k = 0; k < SORT2_PER_THREAD; k++)
    if (a + (uint32_t)k * 1024u + (uint32_t)t < b)
      stage[atomicAdd(&cnt[e[k] >> (s2.idxbits + 1)], 1u)] = (e[k] & idxmask) | (((e[k] >> s2.idxbits) & 1u) << 31);
  __syncthreads();
  uint32_t* dst = sorted + (size_t)w * pl.n + a;
  for (uint32_t i = t; i < b - a; i += 1024) dst[i] = stage[i];
}

// The oversized regions: k_sort2_fine_staged appends (window, range) to a work list (oversize[0] = count, cleared by
// k_sort2_scan), and the two slice kernels run a SMALL fixed grid over list x slices - with an empty list (every input whose
// scalars are not skewed) their workgroups read one word and leave.
constexpr int SORT2_FALLBACK_BLOCKS = 128;
// slice s of region (r, w): [a + s * per, a + (s + 1) * per) with per = ceil(len / SLICES)
__device__ __forceinline__ void sort2_slice(const uint32_t* region_start, int R, int w, int r, int s, uint32_t& lo, uint32_t& hi) {
  const uint32_t a = region_start[(size_t)w * (R + 1) + r], b = region_start[(size_t)w * (R + 1) + r + 1];
  const uint32_t per = (b - a + SORT2_SLICES - 1) / SORT2_SLICES;
  lo = min(b, a + (uint32_t)s * per);
  hi = min(b, lo + per);
}
// fcount[((w*64 + r)*SLICES + s) << sh | t]: entries of slice s in bucket t of range r
static __global__ void __launch_bounds__(512) k_sort2_fine_count(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ region_start,
                                                          const uint32_t* __restrict__ oversize, uint32_t* __restrict__ fcount, MsmPlan pl,
                                                          Sort2 s2) {
  __shared__ uint32_t cnt[512];
  const int t = threadIdx.x;
  const uint32_t items = oversize[0] * SORT2_SLICES;
  for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
    const uint32_t wr = oversize[1 + item / SORT2_SLICES];
    const int w = (int)(wr >> 8), r = (int)(wr & 255u), s = (int)(item % SORT2_SLICES);
    uint32_t lo, hi;
    sort2_slice(region_start, 1 << s2.lgr, w, r, s, lo, hi);
    uint32_t base;
    int sh;
    sort2_window(s2, w, base, sh);
    cnt[t] = 0;
    __syncthreads();
    const uint32_t* src = tmp + (size_t)w * pl.n;
    for (uint32_t i0 = lo; i0 < hi; i0 += 4 * 512) {
      uint32_t e[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t i = i0 + k * 512 + t;
        e[k] = i < hi ? src[i] : 0u;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) sort2_tally(cnt, e[k] >> (s2.idxbits + 1), i0 + k * 512 + t < hi);
    }
    __syncthreads();
    if (t < (1 << sh)) fcount[(((((size_t)w << s2.lgr) + r) * SORT2_SLICES + s) << s2.sh) + t] = cnt[t];
    __syncthreads();
  }
}
static __global__ void __launch_bounds__(512) k_sort2_fine_place(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ region_start,
                                                          const uint32_t* __restrict__ oversize, const uint32_t* __restrict__ fcount,
                                                          uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ sorted, MsmPlan pl,
                                                          Sort2 s2) {
  __shared__ uint32_t cur[512];
  __shared__ uint32_t scan[512];
  const int t = threadIdx.x;
  const uint32_t items = oversize[0] * SORT2_SLICES;
  for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
    const uint32_t wr = oversize[1 + item / SORT2_SLICES];
    const int w = (int)(wr >> 8), r = (int)(wr & 255u), s = (int)(item % SORT2_SLICES);
    uint32_t lo, hi;
    sort2_slice(region_start, 1 << s2.lgr, w, r, s, lo, hi);
    uint32_t base;
    int sh;
    sort2_window(s2, w, base, sh);
    // bucket sizes = sums over the slices -> the range's bucket starts (written by slice 0); this slice's cursors = the start
    // plus what the earlier slices hold of the bucket (every slice of a region redoes the 512-wide scan: cheaper than a launch)
    const int BL = 1 << sh;
    const uint32_t* fc = fcount + ((((((size_t)w << s2.lgr) + r) * SORT2_SLICES)) << s2.sh) + t;
    uint32_t v = 0, before = 0;
#pragma unroll
    for (int k = 0; k < SORT2_SLICES; k++) {
      const uint32_t c = t < BL ? fc[(size_t)k << s2.sh] : 0u;
      v += c;
      if (k < s) before += c;
    }
    const uint32_t a = region_start[(size_t)w * ((1 << s2.lgr) + 1) + r];
    const uint32_t start = sort2_starts(scan, v, a, r, w, BL, base, 1 << s2.lgr, region_start, bucket_start, pl, s == 0);
    cur[t] = start + before;
    __syncthreads();
    const uint32_t* src = tmp + (size_t)w * pl.n;
    uint32_t* dst = sorted + (size_t)w * pl.n;
    const uint32_t idxmask = (1u << s2.idxbits) - 1u;
    for (uint32_t i0 = lo; i0 < hi; i0 += 4 * 512) {
      uint32_t e[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t i = i0 + k * 512 + t;
        e[k] = i < hi ? src[i] : 0u;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool valid = i0 + k * 512 + t < hi;
        const uint32_t pos = sort2_claim(cur, e[k] >> (s2.idxbits + 1), valid);
        if (valid) dst[pos] = (e[k] & idxmask) | (((e[k] >> s2.idxbits) & 1u) << 31);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ 3c. the whole sort of a SMALL plan in one launch (round 6)
// Up to 2^15 points the six sort launches (digits, hist, totals, scan, scatter x 1-2) are 4-7 us kernels 8-10 us apart: 50-65 us of a
// 0.45 ms device phase that does 50 us of arithmetic (profiles/r06_msm_small_timeline.txt).  Here ONE workgroup per window does all
// of it: the window's digit of every scalar (cut from k + H' exactly as k_msm_digits does, kept as int16 in LDS), the bucket
// histogram, its exclusive scan (= bucket_start[w][0..nb]) and the placement into bucket order, with LDS atomics throughout.
// The same workgroup clears its window's bucket accumulators (the accumulate kernel writes non-empty buckets only) and block 0 the
// fix-up's work-list counter, which saves the two fill launches as well.  Same outputs as the separate kernels (the order inside
// a bucket is arbitrary there too: LDS atomics).
constexpr int MSM_SMALL_N = 1 << 15;     // points: 64 KB of LDS digits
constexpr int MSM_SMALL_NB = 1 << 13;    // buckets per window: 32 KB of LDS counters
static __global__ void __launch_bounds__(1024) k_msm_sort_small(const uint32_t* __restrict__ scalars, uint32_t* __restrict__ bucket_start,
                                                         uint32_t* __restrict__ sorted, uint32_t* __restrict__ bad_index,
                                                         uint32_t* __restrict__ buckets, int acc_words, uint32_t* __restrict__ long_runs,
                                                         MsmPlan pl) {
#ifdef __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) uint32_t small_lds[];   // nb counters, then n int16 digits
  __shared__ uint32_t wtot[16];
  uint32_t* hist = small_lds;
  int16_t* dcache = reinterpret_cast<int16_t*>(small_lds + pl.nb);
  const int w = blockIdx.x, t = threadIdx.x;
  for (int b = t; b < pl.nb; b += 1024) hist[b] = 0;
  if (w == 0 && t < 4) long_runs[t] = 0;
  {  // this window's bucket accumulators: all-zero words decode as the identity
    uint4* bz = reinterpret_cast<uint4*>(buckets + (size_t)w * pl.nb * acc_words);
    const int quads = (pl.nb * acc_words) >> 2;
    for (int i = t; i < quads; i += 1024) bz[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  const uint32_t mask = (1u << pl.c) - 1u;
  const int half = 1 << (pl.c - 1);
  const int nwt = pl.nwin_total ? pl.nwin_total : pl.nwin;
  const int bp = (w + pl.w0) * pl.c, limb = bp >> 5, sft = bp & 31;
  const bool top_spread = pl.top_tb && w + pl.w0 == nwt - 1;
  for (int i0 = 0; i0 < pl.n; i0 += 1024) {   // whole waves go round together (sort2_tally)
    const int i = i0 + t;
    int dg = 0;
    if (i < pl.n) {
      uint32_t k[8];
      const uint4* kp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
      const uint4 lo = kp[0], hi = kp[1];
      k[0] = lo.x; k[1] = lo.y; k[2] = lo.z; k[3] = lo.w;
      k[4] = hi.x; k[5] = hi.y; k[6] = hi.z; k[7] = hi.w;
      if (w == 0) {   // one workgroup checks the range of the scalars
        uint32_t bw = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) (void)__builtin_subc(k[j], pl.order[j], bw, &bw);
        if (bw == 0) atomicMin(bad_index, pl.index_base + (uint32_t)i);  // scalar >= order
      }
      uint32_t sum[11];
      uint32_t cy = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) sum[j] = __builtin_addc(k[j], pl.hconst[j], cy, &cy);
      sum[8] = __builtin_addc(0u, pl.hconst[8], cy, &cy);
      sum[9] = pl.hconst[9] + cy;
      sum[10] = 0;
      uint32_t lo32 = 0, hi32 = 0;   // limbs `limb`, `limb + 1` without indexing the register array dynamically
#pragma unroll
      for (int j = 0; j < 10; j++) {
        if (j == limb) { lo32 = sum[j]; hi32 = sum[j + 1]; }
      }
      const uint64_t two = ((uint64_t)hi32 << 32) | lo32;
      const uint32_t v = (uint32_t)(two >> sft) & mask;
      dg = (int)v - half;
      if (top_spread && dg > 0)   // short top window: sub-bucket by the point index (MsmPlan::top_tb)
        dg = (int)((((uint32_t)i & pl.top_submask) << pl.top_tb) + (uint32_t)min(dg, 1 << pl.top_tb));
      dcache[i] = (int16_t)dg;
    }
    sort2_tally(hist, (uint32_t)((dg < 0 ? -dg : dg) - 1), dg != 0);
  }
  __syncthreads();
  // exclusive scan of the nb sizes: thread t owns a run of `per` buckets
  {
    const int per = (pl.nb + 1023) >> 10;
    const int b0 = min(pl.nb, t * per), b1 = min(pl.nb, b0 + per);
    uint32_t mine = 0;
    for (int b = b0; b < b1; b++) mine += hist[b];
    uint32_t incl = mine;
    const int lane = t & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= off) incl += up;
    }
    if (lane == 63) wtot[t >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int j = 0; j < (t >> 6); j++) base += wtot[j];
    uint32_t run = base + incl - mine;
    uint32_t* bs = bucket_start + (size_t)w * (pl.nb + 1);
    for (int b = b0; b < b1; b++) {
      const uint32_t size = hist[b];
      hist[b] = run;   // the bucket's cursor
      bs[b] = run;
      run += size;
    }
    if (t == 1023) bs[pl.nb] = base + incl;
  }
  __syncthreads();
  uint32_t* dst = sorted + (size_t)w * pl.n;
  for (int i0 = 0; i0 < pl.n; i0 += 1024) {
    const int i = i0 + t;
    const int dg = i < pl.n ? (int)dcache[i] : 0;
    const bool valid = dg != 0;
    const uint32_t pos = sort2_claim(hist, (uint32_t)((dg < 0 ? -dg : dg) - 1), valid);
    if (valid) dst[pos] = (uint32_t)i | (dg < 0 ? 0x80000000u : 0u);
  }
#endif
}
static bool msm_small_sort_ok(const MsmPlan& pl) {
  static const int on = knob("NCG_MSM_SMALL_SORT", 1);   // A/B builds: 0 = the separate kernels
  return on && !pl.endo && !pl.shared && pl.part_flags == 3 && pl.n_layout <= pl.n && pl.n <= MSM_SMALL_N && pl.nb <= MSM_SMALL_NB;
}

// does the two-level form apply to this plan?  Fills the per-plan constants.
static bool msm_sort2_ok(const MsmPlan& pl, int n_max, Sort2* s2) {
  static const int on = knob("NCG_MSM_SORT2", 1);   // A/B builds: 0 = the one-level kernels
  if (!on || pl.shared || pl.nb < 1024 || pl.nb > (1 << 15)) return false;
  int lg = 0;
  while ((1 << lg) < pl.nb) lg++;
  int ib = 1;
  while (ib < 31 && (1u << ib) < (unsigned)std::max(n_max, 2)) ib++;
  // 64 ranges per window; 128 / 256 where a region would otherwise hold more than 16 384 entries on average (the staged kernel
  // takes up to SORT2_STAGE = 20 480): the 2^21 entries of an endomorphism plan over 2^20 G1 points, plans of 2^21 / 2^22 points
  int lgr = 6;
  while (lgr < 8 && (n_max >> lgr) > 16384 && lg - lgr > 1) lgr++;
  if (ib + 1 + (lg - lgr) > 32) return false;
  s2->lgr = lgr;
  s2->sh = lg - lgr;
  s2->idxbits = ib;
  s2->top_w = -1;
  s2->top_base = 0;
  s2->top_sh = lg - lgr;
  if (!pl.endo && !pl.top_tb) {   // generic plans whose top window is not spread (MsmPlan::top_tb): the top window's field v = (k + H') >> c (nwin - 1) lies in [half, vmax] (H' carries the window's
                    // own half), so its digits v - half are >= 0 and its buckets are [0, vmax - half) only
    const int nwt = pl.nwin_total ? pl.nwin_total : pl.nwin;
    const int wl = nwt - 1 - pl.w0;   // local index of the top window
    if (wl >= 0 && wl < pl.nwin) {
      // the ONE computation of the top field's bound (msm_plan.hpp; ADVICE r05: the sort, the spread and the tail must agree on it).
      // A scalar >= the group order can exceed it: its top digit then wraps through `& (R - 1)` into another region of this window -
      // memory-safe (every region index stays below R), and harmless only because such a call returns the bad-scalar error and
      // its sum is discarded by every caller (k_msm_digits records the index; msm_finish_t / job_finish_host check it first)
      const uint32_t vmax = msm_plan_top_vmax(pl);
      const uint32_t half = 1u << (pl.c - 1);
      const uint32_t span = vmax >= half ? vmax - half + 2u : half;   // buckets the window can reach: largest digit vmax - half, + 1, + 1 of slack (msm_plan_top_spread: maxd + 1)
      int tb = 0;
      while ((1u << tb) < span) tb++;
      if ((1u << tb) < half) {
        s2->top_w = wl;
        s2->top_base = 0;
        s2->top_sh = std::max(0, tb - lgr);
      }
    }
  }
  return true;
}
// words of the scratch the two-level sort keeps where the one-level sort keeps its per-chunk bucket counts
static size_t msm_sort2_words(const MsmPlan& pl) {
  return (size_t)pl.nwin * ((size_t)pl.Q * SORT2_MAX_RANGES + 2 * (SORT2_MAX_RANGES + 1) + 3) + 8 + (size_t)pl.nwin * pl.nb * SORT2_SLICES;
}

// ------------------------------------------------------------------ 4. bucket accumulation
// Balanced, segmented: every lane owns exactly SEG consecutive entries of a window's sorted
// list (so all lanes of a wave run the same number of mixed adds no matter how the scalars are
// distributed - the reference's own benchmark feeds identical scalars, benchmark/bls12-381.ts:
// 64-79).  A bucket that lies wholly inside a lane's range is written straight to
// buckets[w][b]; a bucket cut by a range boundary leaves partial sums (at most two per lane:
// "head" = the bucket began before the range, "tail" = it continues past the range), which
// k_msm_fixup adds up.
// (MsmSeg / msm_seg: msm_plan.hpp - host-side planning, shared with the CPU test twin)

#ifndef NCG_ACCUM_MINW
#define NCG_ACCUM_MINW 1
#endif
// lane-paired G2 sits 24 registers above the 2-waves/SIMD line: ask for 2 (a little scratch
// traffic in a loop of ~17k instructions per add is cheaper than half the occupancy)
template <class C> struct AccumMinWaves { static constexpr int value = NCG_ACCUM_MINW; };
// G2: the row-wise Montgomery product needed 306 registers (one wave per SIMD; at two waves the kernel kept 79 values in
// scratch memory: 3 % slower on most boxes and 49 % slower on the boxes of the pool whose memory path is slow,
// profiles/r03_box_to_box.md).  With the column-wise product (fp29.hpp mont_cols29) it needs 232: two waves, no scratch.
template <> struct AccumMinWaves<CurveG2P> { static constexpr int value = NCG_G2_ACCUM_WAVES; };
template <> struct AccumMinWaves<CurveG1> { static constexpr int value = 2; };  // 256 VGPRs + 36 B scratch: 2 % faster than 1 wave
// Point prefetch through LDS-DMA (gfx950 global_load_lds_dwordx4: global -> LDS with no register in between).  One mixed
// addition is ~9-10 k instructions (~20 us per lane at two waves per SIMD) and begins with a dependent pair of loads - the
// sorted entry, then the 112 / 224-byte point it names, a random gather - that only one other wave can cover.  With the
// prefetch the point of entry pos + 1 travels into the wave's LDS slab while the addition of entry pos runs, and the entry
// of pos + 2 into a register; the addition starts from 7 ds_read_b128.  Holding the next point in registers instead would
// cost 28 VGPRs that neither kernel has (256 / 232 in use).  Layout of a wave's slab: chunk j (16 bytes per lane) of every
// lane at [j][lane] - the only layout the instruction writes (LDS address = uniform base + lane * 16).  G1: a lane's seven
// chunks are its point (x, y).  Lane-paired G2: the even lane fetches x (c0, c1), the odd lane y (c0, c1), 112 bytes each,
// and each lane then reads its own component of both coordinates from its own and its partner's chunks.
#ifndef NCG_ACCUM_PREFETCH
#define NCG_ACCUM_PREFETCH 0
#endif
template <class C> struct AccumPrefetch { static constexpr bool value = false; };
template <> struct AccumPrefetch<CurveG1> { static constexpr bool value = NCG_ACCUM_PREFETCH != 0; };
template <> struct AccumPrefetch<CurveG2P> { static constexpr bool value = NCG_ACCUM_PREFETCH != 0; };
constexpr int ACCUM_PF_CHUNKS = 7;                          // 7 x 16 bytes per lane
[[maybe_unused]] constexpr int ACCUM_PF_WAVE_WORDS = ACCUM_PF_CHUNKS * 64 * 4;  // one wave's slab, in words

// this lane's 112 bytes starting at `src` (16-byte aligned) -> chunk slots of the wave's slab
NCG_DI void accum_pf_issue(const uint32_t* src, uint32_t* slab) {
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
  for (int j = 0; j < ACCUM_PF_CHUNKS; j++)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * j),
                                     (__attribute__((address_space(3))) void*)(slab + j * 256), 16, 0, 0);
#endif
}
// word k (0..27) of the 112 bytes that lane `ln` of this wave fetched
NCG_DI uint32_t accum_pf_word(const uint32_t* slab, int ln, int k) { return slab[(k >> 2) * 256 + ln * 4 + (k & 3)]; }

template <class C> struct AccumPf;
template <> struct AccumPf<CurveG1> {
  using G = MsmGroup<CurveG1>;
  static NCG_DI const uint32_t* src(const uint32_t* point) { return point; }
  static NCG_DI typename G::Aff read(const uint32_t* slab, int ln) {
    typename G::Aff q;
#pragma unroll
    for (int i = 0; i < 14; i++) {
      q.x.v[i] = accum_pf_word(slab, ln, i);
      q.y.v[i] = accum_pf_word(slab, ln, 14 + i);
    }
    return q;
  }
};
template <> struct AccumPf<CurveG2P> {
  using G = MsmGroup<CurveG2P>;
  static NCG_DI const uint32_t* src(const uint32_t* point) { return point + (pair_odd() ? 28 : 0); }
  static NCG_DI typename G::Aff read(const uint32_t* slab, int ln) {
    const int odd = ln & 1, even_ln = ln & ~1, odd_ln = ln | 1;
    typename G::Aff q;
#pragma unroll
    for (int i = 0; i < 14; i++) {
      q.x.h.v[i] = odd ? accum_pf_word(slab, even_ln, 14 + i) : accum_pf_word(slab, even_ln, i);
      q.y.h.v[i] = odd ? accum_pf_word(slab, odd_ln, 14 + i) : accum_pf_word(slab, odd_ln, i);
    }
    return q;
  }
};

// next non-empty bucket after b (the one that holds sorted position `pos`).  Dense windows: the neighbour, found by walking.
// SPARSE windows (a spread top window that few scalars reach: MsmPlan::top_tb) would walk thousands of empty buckets one dependent
// load at a time - 0.4 ms for EIGHT entries in a 16 384-bucket window - so there a second empty bucket sends the search to a
// bisection.  The two forms are two instantiations of the kernel: the accumulate loop is register-bound (256 VGPRs + spills on
// G1) and ANY extra instruction in it moves its code generation - the bisection in the loop, inline or out of line, cost the
// 2^20-point G1 MSM 2-3 % (3.25 -> 3.37 ms, tools/ab_two_libs.sh), and that plan has no sparse window.
__device__ __noinline__ int msm_bisect_bucket(const uint32_t* __restrict__ bs, int b, uint32_t pos, int nb) {
  int l = b, r = nb;  // bs[l] <= pos < bs[r]
  while (r - l > 1) {
    const int m = (l + r) >> 1;
    if (bs[m] <= pos) l = m; else r = m;
  }
  return l;
}
template <bool SPARSE>
__device__ __forceinline__ int msm_next_bucket(const uint32_t* __restrict__ bs, int b, uint32_t pos, int nb) {
  if constexpr (SPARSE) {
    b++;
    if (bs[b + 1] <= pos) b = msm_bisect_bucket(bs, b, pos, nb);
    return b;
  } else {
    do { b++; } while (bs[b + 1] <= pos);
    return b;
  }
}

template <class C, bool SPARSE>
__global__ void __launch_bounds__(256, AccumMinWaves<C>::value) k_msm_accum(const uint32_t* __restrict__ pts_mont,
                                                   const uint32_t* __restrict__ sorted,
                                                   const uint32_t* __restrict__ bucket_start,
                                                   uint32_t* __restrict__ buckets, uint32_t* __restrict__ part_pts,
                                                   int* __restrict__ part_meta, MsmPlan pl, MsmSeg sg) {
  using G = MsmGroup<C>;
  using Acc = typename G::Acc;
  constexpr int AFF = G::AFF_WORDS, XW = G::ACC_WORDS;
  const int s = (blockIdx.x * blockDim.x + threadIdx.x) >> LaneShift<C>::value, w = blockIdx.y;
  if (s >= sg.nseg) return;
  const uint32_t* bs = bucket_start + (size_t)w * (pl.nb + 1);
  const uint32_t total = bs[pl.nb];
  int* meta = part_meta + ((size_t)w * sg.nseg + s) * 4;  // head bucket, head continues, tail bucket, -
  const uint32_t lo = (uint32_t)s * sg.seg;
  if (lo >= total) {
    meta[0] = -1;
    meta[2] = -1;
    return;
  }
  const uint32_t hi = min(total, lo + (uint32_t)sg.seg);
  // bucket containing position lo: largest b with bs[b] <= lo  (then bs[b+1] > lo)
  int b;
  {
    int l = 0, r = pl.nb;  // invariant: bs[l] <= lo < bs[r]
    while (r - l > 1) {
      int m = (l + r) >> 1;
      if (bs[m] <= lo) l = m; else r = m;
    }
    b = l;
  }
  uint32_t b_start = bs[b], b_end = bs[b + 1];
  const uint32_t* sw = sorted + (size_t)w * pl.n;
  uint32_t* hp = part_pts + (((size_t)w * sg.nseg + s) * 2) * XW;
  int head_b = -1, head_cont = 0, tail_b = -1;
  // a later part of a multi-part MSM (MsmPlan::part_flags): the lane where a bucket STARTS continues from the sum the
  // earlier parts left in that bucket; every other piece starts from the identity as before
  const bool into = (pl.part_flags & 1) == 0;
  Acc acc = (into && b_start >= lo) ? G::acc_load(buckets + ((size_t)w * pl.nb + b) * XW) : G::identity();
  if constexpr (AccumPrefetch<C>::value) {
#ifdef __HIP_DEVICE_COMPILE__
    __shared__ __attribute__((aligned(16))) uint32_t pf_lds[4 * ACCUM_PF_WAVE_WORDS];
    uint32_t* slab = pf_lds + (threadIdx.x >> 6) * ACCUM_PF_WAVE_WORDS;
    const int ln = threadIdx.x & 63;
    uint32_t e_cur = sw[lo];
    accum_pf_issue(AccumPf<C>::src(pts_mont + (size_t)(e_cur & 0x7fffffffu) * AFF), slab);
    uint32_t e_next = lo + 1 < hi ? sw[lo + 1] : e_cur;
    for (uint32_t pos = lo; pos < hi; pos++) {
      // the point of entry pos has landed (the wave's own vmcnt is what orders its ds_read behind the DMA)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const typename G::Aff q = AccumPf<C>::read(slab, ln);
      const bool neg = (e_cur >> 31) != 0;
      // the slab is free again once the reads have returned: send for the next point, and for the entry after it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (pos + 1 < hi) accum_pf_issue(AccumPf<C>::src(pts_mont + (size_t)(e_next & 0x7fffffffu) * AFF), slab);
      e_cur = e_next;
      if (pos + 2 < hi) e_next = sw[pos + 2];
      if (pos == b_end) {  // bucket finished inside my range
        if (b_start >= lo) {
          G::acc_store(buckets + ((size_t)w * pl.nb + b) * XW, acc);
        } else {
          G::acc_store(hp, acc);
          head_b = b;
        }
        b = msm_next_bucket<SPARSE>(bs, b, pos, pl.nb);
        b_start = bs[b];
        b_end = bs[b + 1];
        acc = into ? G::acc_load(buckets + ((size_t)w * pl.nb + b) * XW) : G::identity();
      }
      acc = G::madd(acc, q, neg);
    }
#endif
  } else
  for (uint32_t pos = lo; pos < hi; pos++) {
    if (pos == b_end) {  // bucket finished inside my range
      if (b_start >= lo) {
        G::acc_store(buckets + ((size_t)w * pl.nb + b) * XW, acc);
      } else {
        G::acc_store(hp, acc);
        head_b = b;
      }
      b = msm_next_bucket<SPARSE>(bs, b, pos, pl.nb);
      b_start = bs[b];
      b_end = bs[b + 1];
      acc = into ? G::acc_load(buckets + ((size_t)w * pl.nb + b) * XW) : G::identity();
    }
    uint32_t e = sw[pos];
    const uint32_t* pp = pts_mont + (size_t)(e & 0x7fffffffu) * AFF;
    acc = G::madd(acc, G::aff_load(pp), (e >> 31) != 0);
  }
  const bool left = b_start >= lo, right = b_end <= hi;
  if (left && right) {
    G::acc_store(buckets + ((size_t)w * pl.nb + b) * XW, acc);
  } else if (!left) {
    G::acc_store(hp, acc);
    head_b = b;
    head_cont = right ? 0 : 1;
  } else {
    G::acc_store(hp + XW, acc);
    tail_b = b;
  }
  meta[0] = head_b;
  meta[1] = head_cont;
  meta[2] = tail_b;
}

// Adds up the pieces of every bucket that was cut by lane boundaries.  The pieces of one bucket
// are the "tail" of the lane where it starts (piece 0) followed by the "head" of every later
// lane it covers; lane s0 = bucket_start / seg and s1 = (bucket_end - 1) / seg delimit the run,
// so every piece knows its place without a scan.  With `seg` entries per lane a bucket of m
// entries is cut into at most m / seg + 2 pieces: for scalars that look random the runs are 2
// (rarely 3) pieces long, so the owner of piece 0 adds the few heads itself and writes the bucket -
// ONE launch (round 2 ran 13-14 log-step passes that all but the first found nothing to do).
// A run longer than MSM_RUN_SERIAL heads (identical scalars: benchmark/bls12-381.ts:64-79 feeds
// them; the short top window of most plans) goes to a work list instead, and k_msm_fixup_long (below, after the
// cooperative operations it uses) gives each such run a whole workgroup: strided partial sums, then an LDS tree -
// log depth in the run length.
constexpr int MSM_RUN_SERIAL = 2;
constexpr int MSM_RUN_SERIAL_SHARED = 8;
constexpr int MSM_LONG_BLOCKS = 512;
template <class C>
__global__ void __launch_bounds__(256, TailMinWaves<C>::value) k_msm_fixup_merge(const uint32_t* __restrict__ part_pts,
                                                        const int* __restrict__ part_meta,
                                                        const uint32_t* __restrict__ bucket_start,
                                                        uint32_t* __restrict__ buckets, MsmPlan pl, MsmSeg sg,
                                                        uint32_t* __restrict__ long_runs, int run_serial) {
  using G = MsmGroup<C>;
  constexpr int XW = G::ACC_WORDS, LS = LaneShift<C>::value;
  const int s = (blockIdx.x * blockDim.x + threadIdx.x) >> LS, w = blockIdx.y;
  if (s >= sg.nseg) return;
  const int tb = part_meta[((size_t)w * sg.nseg + s) * 4 + 2];
  if (tb < 0) return;
  const uint32_t* bs = bucket_start + (size_t)w * (pl.nb + 1);
  const int s1 = (int)((bs[tb + 1] - 1) / (uint32_t)sg.seg);
  const int heads = s1 - s;
  if (heads > run_serial) {
    if ((threadIdx.x & ((1 << LS) - 1)) == 0) {
      const uint32_t k = atomicAdd(long_runs, 1u);
      uint32_t* e = long_runs + 4 + (size_t)k * 4;
      e[0] = (uint32_t)w;
      e[1] = (uint32_t)tb;
      e[2] = (uint32_t)s;
      e[3] = (uint32_t)s1;
    }
    return;
  }
  const uint32_t* pp = part_pts + (size_t)w * sg.nseg * 2 * XW;
  typename G::Acc acc = G::acc_load(pp + ((size_t)s * 2 + 1) * XW);
  for (int k = 1; k <= heads; k++) acc = G::add(acc, G::acc_load(pp + ((size_t)(s + k) * 2) * XW));
  G::acc_store(buckets + ((size_t)w * pl.nb + tb) * XW, acc);
}

// ------------------------------------------------------------------ 5. bucket fold, one level
// in:  [narr][nwin][n_in] XYZZ   (array 0 = S, arrays 1.. = pending R sums)
// out: [narr+1][nwin][n_in/2]    out[a] = pairwise sums (a < narr), out[narr] = odd elements of S
template <class C>
__global__ void __launch_bounds__(256, TailMinWaves<C>::value) k_msm_reduce_level(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                          int narr, int nwin, int n_in) {
  using G = MsmGroup<C>;
  constexpr int XW = G::ACC_WORDS;
  const int n_out = n_in >> 1;
  const long total = (long)(narr + 1) * nwin * n_out;
  const long t = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> LaneShift<C>::value;
  if (t >= total) return;
  const int q = (int)(t % n_out);
  const int w = (int)((t / n_out) % nwin);
  const int a = (int)(t / ((long)n_out * nwin));
  typename G::Acc r;
  if (a < narr) {
    const uint32_t* base = in + (((size_t)a * nwin + w) * n_in + 2 * (size_t)q) * XW;
    r = G::add(G::acc_load(base), G::acc_load(base + XW));
  } else {
    r = G::acc_load(in + (((size_t)w) * n_in + 2 * (size_t)q + 1) * XW);
  }
  G::acc_store(out + (((size_t)a * nwin + w) * n_out + q) * XW, r);
}

// ------------------------------------------------------------------ 5a. the same level, four items per addition
// For the levels that no longer fill the chip (msm_coop.hpp): same indexing, one task per GROUP of four items.
template <class C>
__global__ void __launch_bounds__(256) k_msm_reduce_level_coop(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                               int narr, int nwin, int n_in) {
#ifdef __HIP_DEVICE_COMPILE__
  using K = CoopXyzz<C>;
  constexpr int XW = MsmGroup<C>::ACC_WORDS;
  extern __shared__ __attribute__((aligned(16))) uint32_t coop_lds[];
  uint32_t* lds = coop_lds + (size_t)K::group_in_block() * K::GROUP_WORDS;
  const int n_out = n_in >> 1;
  const long total = (long)(narr + 1) * nwin * n_out;
  const long t = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> (K::LS + 2);
  if (t >= total) return;
  const int q = (int)(t % n_out);
  const int w = (int)((t / n_out) % nwin);
  const int a = (int)(t / ((long)n_out * nwin));
  uint32_t* dst = out + (((size_t)a * nwin + w) * n_out + q) * XW;
  if (a < narr) {
    const uint32_t* base = in + (((size_t)a * nwin + w) * n_in + 2 * (size_t)q) * XW;
    K::add(lds, base, base + XW, dst);
  } else {
    K::copy(in + (((size_t)w) * n_in + 2 * (size_t)q + 1) * XW, dst);
  }
#endif
}

// ------------------------------------------------------------------ 5b. the narrow end in ONE launch
// One workgroup per window runs the last fold levels (from the point where a level's additions fit the
// workgroup once or twice) and then groups the pending sums, with workgroup barriers instead of launches:
// After the fold every window holds narr = c points: array 0 = S (weight 1) and array a >= 1 =
// the level-a pending sum (weight 2^(a-1)), i.e. terms u_e = at(e+1) (+ at(0) for e = 0) with
// weight 2^e, e < c-1.  The host Horner would spend one addition per term; here g consecutive
// terms are pre-combined, V_j = sum_{i<g} 2^i u_{jg+i} (g-1 doublings + additions per unit, all
// groups in parallel), so the host does one addition per GROUP - its doublings (one per scalar
// bit) are the part only a latency-optimised core can do quickly.
// TailOps<C, COOP>: a "unit" is one item (lane / lane pair) running the complete single-lane routines, or a
// group of four items sharing each operation (msm_coop.hpp).  Operands and results live in memory; `lds` is the
// unit's scratch: the exchange slots of the cooperative form, then one accumulator for the grouping chain.
constexpr int MSM_TAIL_THREADS = 512;
#ifndef NCG_TAIL_ROUNDS
#define NCG_TAIL_ROUNDS 1
#endif
template <class C, bool COOP> struct TailOps;
template <class C>
struct TailOps<C, false> {
  using G = MsmGroup<C>;
  static constexpr int UNIT_SHIFT = LaneShift<C>::value;
  static constexpr int SCRATCH_WORDS = 0;
  static constexpr int LDS_WORDS = G::ACC_WORDS;  // the chain accumulator
  static __device__ __noinline__ void add(uint32_t* lds, const uint32_t* p, const uint32_t* q, uint32_t* out) {
    G::acc_store(out, G::add(G::acc_load(p), G::acc_load(q)));
  }
  static __device__ __noinline__ void dbl(uint32_t* lds, const uint32_t* p, uint32_t* out) { G::acc_store(out, G::dbl(G::acc_load(p))); }
  static __device__ __forceinline__ void copy(const uint32_t* p, uint32_t* out) { G::acc_store(out, G::acc_load(p)); }
  static __device__ __forceinline__ void sync() {}
};
#ifdef __HIP_DEVICE_COMPILE__
template <class C>
struct TailOps<C, true> {
  using K = CoopXyzz<C>;
  static constexpr int UNIT_SHIFT = LaneShift<C>::value + 2;
  static constexpr int SCRATCH_WORDS = K::GROUP_WORDS;
  static constexpr int LDS_WORDS = K::GROUP_WORDS + MsmGroup<C>::ACC_WORDS;
  static __device__ __noinline__ void add(uint32_t* lds, const uint32_t* p, const uint32_t* q, uint32_t* out) { K::add(lds, p, q, out); }
  static __device__ __noinline__ void dbl(uint32_t* lds, const uint32_t* p, uint32_t* out) { K::dbl(lds, p, out); }
  static __device__ __forceinline__ void copy(const uint32_t* p, uint32_t* out) { K::copy(p, out); }
  static __device__ __forceinline__ void sync() { coop_sync(); }
};
#else
template <class C>
struct TailOps<C, true> : TailOps<C, false> {
  static constexpr int UNIT_SHIFT = LaneShift<C>::value + 2;
  static constexpr int SCRATCH_WORDS = COOP_SLOTS * MsmGroup<C>::FW;
  static constexpr int LDS_WORDS = COOP_SLOTS * MsmGroup<C>::FW + MsmGroup<C>::ACC_WORDS;
};
#endif

// Long runs of the fix-up (see k_msm_fixup_merge): one workgroup per run of the work list.  Unit t sums the pieces
// t, t + act, ... (act = the power of two >= the run length, at most the workgroup's units), then a tree over the
// units' accumulators in LDS.  Units are cooperative groups where the curve offers them: a run of 50 pieces (the
// short top window of a 2^17-point shard) costs 1 + 6 cooperative additions.
template <class C, bool COOP>
__global__ void __launch_bounds__(MSM_TAIL_THREADS) k_msm_fixup_long(const uint32_t* __restrict__ part_pts,
                                                                     uint32_t* __restrict__ buckets, MsmPlan pl, MsmSeg sg,
                                                                     const uint32_t* __restrict__ long_runs) {
#ifdef __HIP_DEVICE_COMPILE__
  using T = TailOps<C, COOP>;
  constexpr int XW = MsmGroup<C>::ACC_WORDS;
  extern __shared__ __attribute__((aligned(16))) uint32_t long_lds[];
  const uint32_t count = long_runs[0];
  if (count == 0) return;  // the usual case for full top windows: one wave-uniform load and out
  const int unit = (int)(threadIdx.x >> T::UNIT_SHIFT), units = (int)(blockDim.x >> T::UNIT_SHIFT);
  const int lane_in_unit = (int)(threadIdx.x & ((1u << T::UNIT_SHIFT) - 1u));
  uint32_t* lds = long_lds + (size_t)unit * T::LDS_WORDS;
  uint32_t* acc = lds + T::SCRATCH_WORDS;
  for (uint32_t r = blockIdx.x; r < count; r += gridDim.x) {
    const uint32_t* e = long_runs + 4 + (size_t)r * 4;
    const int w = (int)e[0], tb = (int)e[1], s0 = (int)e[2], s1 = (int)e[3];
    const uint32_t* pp = part_pts + (size_t)w * sg.nseg * 2 * XW;
    const int np = s1 - s0 + 1;  // piece 0 = the tail slot of lane s0, piece k >= 1 = the head slot of lane s0 + k
    auto piece = [&](int k) { return pp + ((size_t)(s0 + k) * 2 + (k == 0 ? 1 : 0)) * XW; };
    int act = units;
    while ((act >> 1) >= np) act >>= 1;
    if (unit < act) {
      if (unit < np) {
        T::copy(piece(unit), acc);
      } else {
        for (int i = lane_in_unit; i < XW; i += (1 << T::UNIT_SHIFT)) acc[i] = 0;  // identity
      }
      T::sync();
      for (int k = unit + act; k < np; k += act) {
        T::add(lds, acc, piece(k), acc);
        T::sync();
      }
    }
    for (int off = act >> 1; off >= 1; off >>= 1) {
      __syncthreads();
      if (unit < off) T::add(lds, acc, long_lds + (size_t)(unit + off) * T::LDS_WORDS + T::SCRATCH_WORDS, acc);
    }
    __syncthreads();
    if (unit == 0) T::copy(acc, buckets + ((size_t)w * pl.nb + tb) * XW);
    __syncthreads();
  }
#endif
}

// The fix-up merge with one UNIT (cooperative group) per lane segment, for the shared-bucket mode: there every bucket is
// cut into a handful of pieces (nwin * n / nb entries over lanes of `seg`), the owners are one lane in four or five and
// each adds 4-5 heads - a dependent chain that the cooperative additions shorten 2-3x.  Same contract as
// k_msm_fixup_merge (runs longer than run_serial heads go to the work list).
template <class C, bool COOP>
__global__ void __launch_bounds__(256, 2) k_msm_fixup_merge_units(const uint32_t* __restrict__ part_pts, const int* __restrict__ part_meta,
                                                               const uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ buckets,
                                                               MsmPlan pl, MsmSeg sg, uint32_t* __restrict__ long_runs, int run_serial) {
#ifdef __HIP_DEVICE_COMPILE__
  using T = TailOps<C, COOP>;
  constexpr int XW = MsmGroup<C>::ACC_WORDS;
  extern __shared__ __attribute__((aligned(16))) uint32_t merge_lds[];
  const int unit = (int)(threadIdx.x >> T::UNIT_SHIFT);
  uint32_t* lds = merge_lds + (size_t)unit * T::LDS_WORDS;
  uint32_t* acc = lds + T::SCRATCH_WORDS;
  // one unit per BUCKET (the owners are one lane in four or five: indexing by lane would spread them over every
  // workgroup): the run of bucket tb starts in lane s = start / seg, whose tail slot names tb iff the bucket is cut
  const int tb = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> T::UNIT_SHIFT), w = blockIdx.y;
  if (tb >= pl.nb) return;
  const uint32_t* bs = bucket_start + (size_t)w * (pl.nb + 1);
  const uint32_t b0 = bs[tb], b1 = bs[tb + 1];
  if (b1 == b0) return;
  const int s = (int)(b0 / (uint32_t)sg.seg);
  if (part_meta[((size_t)w * sg.nseg + s) * 4 + 2] != tb) return;
  const int s1 = (int)((b1 - 1) / (uint32_t)sg.seg);
  const int heads = s1 - s;
  if (heads > run_serial) {
    if ((threadIdx.x & ((1u << T::UNIT_SHIFT) - 1u)) == 0) {
      const uint32_t k = atomicAdd(long_runs, 1u);
      uint32_t* e = long_runs + 4 + (size_t)k * 4;
      e[0] = (uint32_t)w;
      e[1] = (uint32_t)tb;
      e[2] = (uint32_t)s;
      e[3] = (uint32_t)s1;
    }
    return;
  }
  const uint32_t* pp = part_pts + (size_t)w * sg.nseg * 2 * XW;
  uint32_t* dst = buckets + ((size_t)w * pl.nb + tb) * XW;
  if (heads == 0) {
    T::copy(pp + ((size_t)s * 2 + 1) * XW, dst);
    return;
  }
  T::copy(pp + ((size_t)s * 2 + 1) * XW, acc);
  T::sync();
  for (int k = 1; k < heads; k++) {
    T::add(lds, acc, pp + ((size_t)(s + k) * 2) * XW, acc);
    T::sync();
  }
  T::add(lds, acc, pp + ((size_t)(s + heads) * 2) * XW, dst);
#endif
}

// The fix-up merge of the SMALL plans (round 6): 2^MERGE_TREE_ULOG units per bucket (two: measured against four and against one
// unit with the work-list threshold at 15, tools/ab_small_msm.sh -> profiles/r06_ab_small_msm.txt: four units idle through most of
// the tree and quadruple the waves - 83 us for the 2^14-point G1 merge where two take 60).  A plan that does not fill the chip runs short lane
// segments (msm_seg: 4-12 entries, so that the accumulate kernel's dependent chain is short), which cuts every bucket into
// (m + 4 sqrt(m)) / seg pieces - up to 8-15 of them - and one unit adding them one after the other made the merge the longest
// kernel of a 2^13 / 2^14-point MSM (62 us, + 35 us of work-list runs above 8 pieces).  Here unit u of the bucket's group sums the
// pieces u, u + U, .. and a tree over the U accumulators (LDS, all in one wave) finishes: ceil(P / U) - 1 + log2 U dependent
// additions for P pieces (two units, P = 15: 8 instead of 14).  Runs of more than run_serial heads still go to the work list.
#ifndef NCG_MERGE_TREE_ULOG
#define NCG_MERGE_TREE_ULOG 1
#endif
constexpr int MERGE_TREE_ULOG = NCG_MERGE_TREE_ULOG;
template <class C, bool COOP>
__global__ void __launch_bounds__(256, 2) k_msm_fixup_merge_tree(const uint32_t* __restrict__ part_pts, const int* __restrict__ part_meta,
                                                              const uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ buckets,
                                                              MsmPlan pl, MsmSeg sg, uint32_t* __restrict__ long_runs, int run_serial) {
#ifdef __HIP_DEVICE_COMPILE__
  using T = TailOps<C, COOP>;
  constexpr int XW = MsmGroup<C>::ACC_WORDS, U = 1 << MERGE_TREE_ULOG;
  extern __shared__ __attribute__((aligned(16))) uint32_t tree_lds[];
  const int unit = (int)(threadIdx.x >> T::UNIT_SHIFT);
  uint32_t* lds = tree_lds + (size_t)unit * T::LDS_WORDS;
  uint32_t* acc = lds + T::SCRATCH_WORDS;
  const int gunit = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> T::UNIT_SHIFT);
  const int tb = gunit >> MERGE_TREE_ULOG, sub = gunit & (U - 1), w = blockIdx.y;
  if (tb >= pl.nb) return;
  const uint32_t* bs = bucket_start + (size_t)w * (pl.nb + 1);
  const uint32_t b0 = bs[tb], b1 = bs[tb + 1];
  if (b1 == b0) return;
  const int s = (int)(b0 / (uint32_t)sg.seg);
  if (part_meta[((size_t)w * sg.nseg + s) * 4 + 2] != tb) return;   // the bucket is not cut
  const int s1 = (int)((b1 - 1) / (uint32_t)sg.seg);
  const int heads = s1 - s;
  if (heads > run_serial) {
    if (sub == 0 && (threadIdx.x & ((1u << T::UNIT_SHIFT) - 1u)) == 0) {
      const uint32_t k = atomicAdd(long_runs, 1u);
      uint32_t* e = long_runs + 4 + (size_t)k * 4;
      e[0] = (uint32_t)w;
      e[1] = (uint32_t)tb;
      e[2] = (uint32_t)s;
      e[3] = (uint32_t)s1;
    }
    return;
  }
  const uint32_t* pp = part_pts + (size_t)w * sg.nseg * 2 * XW;
  uint32_t* dst = buckets + ((size_t)w * pl.nb + tb) * XW;
  const int np = heads + 1;   // piece 0 = the tail slot of lane s, piece k >= 1 = the head slot of lane s + k
  auto piece = [&](int k) { return pp + ((size_t)(s + k) * 2 + (k == 0 ? 1 : 0)) * XW; };
  if (np == 1) {
    if (sub == 0) T::copy(piece(0), dst);
    return;
  }
  // (the units of a bucket sit in one wave: U x 4 lanes on G1, U x 8 on the lane-paired G2; T::sync orders their LDS traffic)
  if (sub < np) {
    T::copy(piece(sub), acc);
    T::sync();
    for (int k = sub + U; k < np; k += U) {
      T::add(lds, acc, piece(k), acc);
      T::sync();
    }
  }
  const int live = np < U ? np : U;   // accumulators that hold something
#pragma unroll
  for (int off = U >> 1; off >= 1; off >>= 1) {
    T::sync();
    if (sub < off && sub + off < live) {
      uint32_t* other = tree_lds + (size_t)(unit + off) * T::LDS_WORDS + T::SCRATCH_WORDS;
      if (off == 1) T::add(lds, acc, other, dst);
      else T::add(lds, acc, other, acc);
    }
  }
#endif
}

// in: [narr][nwin][n_in] accumulators (read-only here: other workgroups read their windows from it);
// s0 / s1: scratch, MSM_TAIL_REGION accumulators PER WINDOW each - a window's levels ping-pong inside its own
// regions, laid out [array][n] (workgroups run at different levels, so they must not share a layout);
// fin: [ngroups][nwin] grouped sums.
constexpr int MSM_TAIL_REGION = 2 * MSM_TAIL_THREADS + 64;
template <class C, bool COOP>
__global__ void __launch_bounds__(MSM_TAIL_THREADS) k_msm_tail(const uint32_t* __restrict__ in, uint32_t* __restrict__ s0,
                                                               uint32_t* __restrict__ s1, uint32_t* __restrict__ fin,
                                                               int narr, int nwin, int n_in, int g, int ngroups, int top_w, int top_tb,
                                                               uint32_t* __restrict__ host_flag, uint32_t host_gen) {
#ifdef __HIP_DEVICE_COMPILE__
  using T = TailOps<C, COOP>;
  if (host_flag && blockIdx.x == 0 && threadIdx.x == 0)   // "the tail has started": the host wakes its finish helpers (msm_finish_t)
    __hip_atomic_store(host_flag, host_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  constexpr int XW = MsmGroup<C>::ACC_WORDS;
  extern __shared__ __attribute__((aligned(16))) uint32_t tail_lds[];
  const int unit = (int)(threadIdx.x >> T::UNIT_SHIFT), units = (int)(blockDim.x >> T::UNIT_SHIFT);
  uint32_t* lds = tail_lds + (size_t)unit * T::LDS_WORDS;
  const int w = blockIdx.x;
  // element (array a, index i) of the current level: the input array is [a][w][i], the private regions [a][i]
  const uint32_t* cur = in;
  size_t cur_a = (size_t)nwin * n_in, cur_w = (size_t)w * n_in;  // strides / offset in accumulators
  uint32_t* nxt = s0 + (size_t)w * MSM_TAIL_REGION * XW;
  uint32_t* spare = s1 + (size_t)w * MSM_TAIL_REGION * XW;
  while (n_in > 1) {
    const int n_out = n_in >> 1;
    const int tasks = (narr + 1) * n_out;
    for (int t = unit; t < tasks; t += units) {
      const int a = t / n_out, q = t - a * n_out;
      uint32_t* dst = nxt + ((size_t)a * n_out + q) * XW;
      if (a < narr) {
        const uint32_t* base = cur + ((size_t)a * cur_a + cur_w + 2 * (size_t)q) * XW;
        T::add(lds, base, base + XW, dst);
      } else {
        T::copy(cur + (cur_w + 2 * (size_t)q + 1) * XW, dst);
      }
    }
    __threadfence_block();
    __syncthreads();
    uint32_t* done = nxt;
    nxt = spare;
    spare = done;
    cur = done;
    cur_a = (size_t)n_out;
    cur_w = 0;
    narr++;
    n_in = n_out;
  }
  if (cur == in) {  // no level ran here (a one-bucket window): array a of this window sits at [a][w]
    cur_a = (size_t)nwin;
    cur_w = (size_t)w;
  }
  // grouping: unit j combines the terms e_lo .. e_hi of this window into fin[j][w]; the chain runs in the unit's LDS accumulator
  uint32_t* acc = lds + T::SCRATCH_WORDS;
  // a spread top window (MsmPlan::top_tb): only the pending sums of the levels below top_tb count - the weight of a bucket
  // is (index mod 2^top_tb) + 1
  const int levels = (w == top_w && top_tb > 0) ? min(top_tb, narr - 1) : narr - 1;
  for (int j = unit; j < ngroups; j += units) {
    const int e_lo = j * g;
    const int e_hi = min((j + 1) * g, levels) - 1;
    if (e_hi < e_lo) {   // nothing of this window in the group: the identity (all-zero accumulators decode as such)
      uint32_t* dst = fin + ((size_t)j * nwin + w) * XW;
      for (int i = (int)(threadIdx.x & ((1u << T::UNIT_SHIFT) - 1u)); i < XW; i += (1 << T::UNIT_SHIFT)) dst[i] = 0;
      continue;
    }
    auto at = [&](int a) { return cur + ((size_t)a * cur_a + cur_w) * XW; };
    T::copy(at(e_hi + 1), acc);
    T::sync();
    if (e_hi == 0) T::add(lds, acc, at(0), acc);
    for (int e = e_hi - 1; e >= e_lo; e--) {
      T::sync();
      T::dbl(lds, acc, acc);
      T::sync();
      T::add(lds, acc, at(e + 1), acc);
      if (e == 0) {
        T::sync();
        T::add(lds, acc, at(0), acc);
      }
    }
    T::sync();
    T::copy(acc, fin + ((size_t)j * nwin + w) * XW);
  }
#endif
}

// ------------------------------------------------------------------ 5c. multi-GPU combine
// in: [nparts][npoints] accumulators (the grouped window sums of nparts shards, all-gathered; scratch - reduced in
// place); out[t] = sum_r in[r][t] - the reference's final `sum.add(resI)` chain over disjoint point ranges
// (src/abstract/curve.ts:895-902 is linear in the points, so partial MSMs add up term by term).
// One workgroup per point, a pairwise tree over the shards (log2(nparts) dependent additions instead of
// nparts - 1; cooperative additions where the group offers them): 8 shards = 3 x ~8 us instead of 7 x ~16 us.
template <class C, bool COOP>
__global__ void __launch_bounds__(256) k_msm_sum_partials(uint32_t* __restrict__ in, uint32_t* __restrict__ out, int nparts,
                                                          int npoints) {
#ifdef __HIP_DEVICE_COMPILE__
  using T = TailOps<C, COOP>;
  constexpr int XW = MsmGroup<C>::ACC_WORDS;
  extern __shared__ __attribute__((aligned(16))) uint32_t sum_lds[];
  const int unit = (int)(threadIdx.x >> T::UNIT_SHIFT), units = (int)(blockDim.x >> T::UNIT_SHIFT);
  uint32_t* lds = sum_lds + (size_t)unit * T::SCRATCH_WORDS;
  const int t = blockIdx.x;
  auto at = [&](int r) { return in + ((size_t)r * npoints + t) * XW; };
  for (int s = 1; s < nparts; s <<= 1) {
    const int pairs = (nparts + 2 * s - 1) / (2 * s);
    for (int k = unit; k < pairs; k += units) {
      const int i = k * 2 * s;
      if (i + s < nparts) T::add(lds, at(i), at(i + s), at(i));
    }
    __threadfence_block();
    __syncthreads();
  }
  if (unit == 0) T::copy(at(0), out + (size_t)t * XW);
#endif
}

// ------------------------------------------------------------------ planning (msm_plan.hpp)
#ifndef NCG_MSM_TU_G1
int msm_make_plan(int curve, int n, int c_override, MsmPlan* pl) { return msm_make_plan_impl(curve, n, c_override, pl); }
#endif

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct MsmLayout {
  size_t pts_mont, digits, counts, bucket_start, sorted, sort_tmp, shared_start, buckets, part_pts, part_meta, long_runs, bad, red0, red1, tail0, tail1, fin, total;
};

template <class C>
static MsmLayout msm_layout(const MsmPlan& pl_in) {
  MsmLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align256(off + bytes);
    return o;
  };
  MsmPlan pl = pl_in;
  if (pl.n_layout > pl.n) pl.n = pl.n_layout;  // parts of one MSM: every part addresses the layout of the largest
  if (pl.Q_layout > pl.Q) pl.Q = pl.Q_layout;
  L.pts_mont = take((pl.endo || pl.pts_stored || pl.shared) ? 0 : (size_t)pl.n * MsmGroup<C>::AFF_WORDS * 4);  // else: the caller's array
  L.digits = take((size_t)pl.nwin * pl.n * 2);
  Sort2 s2l;
  const bool sort2 = msm_sort2_ok(pl, pl.n, &s2l);
  L.counts = take(sort2 ? msm_sort2_words(pl) * 4 : (size_t)pl.nwin * pl.Q * pl.nb * 4);
  L.bucket_start = take((size_t)pl.nwin * (pl.nb + 1) * 4);
  L.sorted = take((size_t)pl.nwin * pl.n * 4);
  L.sort_tmp = take(sort2 ? (size_t)pl.nwin * pl.n * 4 : 0);   // the regions of the two-level sort
  L.shared_start = take((size_t)(pl.nb + 1) * 4);
  const MsmPlan av = msm_acc_view(pl);   // one window of nwin * n entries in shared-bucket mode
  L.buckets = take((size_t)av.nwin * av.nb * MsmGroup<C>::ACC_WORDS * 4);
  MsmSeg sg = msm_seg(av);
  L.part_pts = take((size_t)av.nwin * sg.nseg * 2 * MsmGroup<C>::ACC_WORDS * 4);
  L.part_meta = take((size_t)av.nwin * sg.nseg * 4 * 4);
  // work list of the long runs: a counter + (window, bucket, first lane, last lane) per run of more than
  // MSM_RUN_SERIAL heads - such runs cover disjoint lane ranges, so there are at most nwin * nseg / MSM_RUN_SERIAL
  // (with a run_serial override below 2 - tests - adjacent runs share a lane: at most one run per lane)
  L.long_runs = take(16 + ((size_t)av.nwin * (sg.nseg + 1)) * 16);
  L.bad = take(64);
  // fold ping-pong: level l output holds (l+1) * nwin * nb/2^l points <= nwin*nb (l = 1, 2)
  size_t red = (size_t)av.nwin * std::max(av.nb, av.c) * MsmGroup<C>::ACC_WORDS * 4;
  L.red0 = take(red);
  L.red1 = take(red);
  // the tail workgroups' private ping-pong (levels that fit one workgroup per window: at most 2 * 512 additions
  // per window and level) and the grouped window sums
  const size_t tail = (size_t)av.nwin * MSM_TAIL_REGION * MsmGroup<C>::ACC_WORDS * 4;
  L.tail0 = take(tail);
  L.tail1 = take(tail);
  L.fin = take((size_t)msm_ngroups(av.c) * av.nwin * MsmGroup<C>::ACC_WORDS * 4);
  L.total = off;
  return L;
}

// Device phase: everything up to the grouped window sums (ng x nwin accumulators, device memory,
// inside the workspace).  Asynchronous on `st`.
template <class C>
static hipError_t msm_device_t(const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws,
                               const uint32_t** d_fin, hipStream_t st, const uint32_t** d_bad = nullptr,
                               const MsmSide* side = nullptr) {
  using G = MsmGroup<C>;
  using D = typename DeviceCurve<C>::type;  // kernels: lane-paired form for G2
  constexpr int LS = LaneShift<D>::value;
  constexpr int XW = G::ACC_WORDS;
  MsmLayout L = msm_layout<C>(pl);
  char* base = (char*)ws;
  uint32_t* pts_mont = (uint32_t*)(base + L.pts_mont);
  int16_t* digits = (int16_t*)(base + L.digits);
  uint32_t* counts = (uint32_t*)(base + L.counts);
  uint32_t* bstart = (uint32_t*)(base + L.bucket_start);
  uint32_t* sorted = (uint32_t*)(base + L.sorted);
  uint32_t* buckets = (uint32_t*)(base + L.buckets);
  uint32_t* red[2] = {(uint32_t*)(base + L.red0), (uint32_t*)(base + L.red1)};
  const int n = pl.n;
  hipError_t e;

  const bool part_first = (pl.part_flags & 1) != 0, part_last = (pl.part_flags & 2) != 0;
  if ((!part_first || !part_last) && (pl.shared || pl.endo)) return hipErrorInvalidValue;  // parts: generic plans only
  bool forked = false;
  // error paths between the fork and the join must not return while the side-stream kernel is still writing into the
  // workspace (the caller may free or re-size it): drain the side stream unless the join has been enqueued on `st`
  struct SideGuard {
    const MsmSide* side;
    bool armed = false;
    ~SideGuard() {
      if (armed && side && side->stream) (void)hipStreamSynchronize(side->stream);
    }
  } side_guard{side};
  uint32_t* bad = (uint32_t*)(base + L.bad);
  if (part_first) {
    e = hipMemsetAsync(bad, 0xFF, 4, st);
    if (e != hipSuccess) return e;
  }
  const bool small_sort = msm_small_sort_ok(pl);   // digits + sort + the two clears in one launch (3c)
  if (pl.endo) {  // d_pts is the expanded image set, already in storage format (msm_endo_expand)
    pts_mont = const_cast<uint32_t*>(d_pts);
    e = msm_endo_digits(pl, d_scalars, digits, bad, st);
    if (e != hipSuccess) return e;
  } else {
    if (pl.pts_stored || pl.shared) {
      pts_mont = const_cast<uint32_t*>(d_pts);
    } else if (side && side->stream) {  // beside the digits / sort kernels; joined in front of the accumulate kernel
      e = hipEventRecord(side->fork, st);
      if (e == hipSuccess) e = hipStreamWaitEvent(side->stream, side->fork, 0);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k_points_to_mont<D>, dim3((unsigned)((((size_t)n << LS) + 255) / 256)), dim3(256), 0, side->stream,
                         d_pts, pts_mont, n);
      side_guard.armed = true;
      e = hipEventRecord(side->join, side->stream);
      if (e != hipSuccess) return e;
      forked = true;
    } else {
      hipLaunchKernelGGL(k_points_to_mont<D>, dim3((unsigned)((((size_t)n << LS) + 255) / 256)), dim3(256), 0, st, d_pts,
                         pts_mont, n);
    }
    if (!small_sort) hipLaunchKernelGGL(k_msm_digits, dim3((n + 255) / 256), dim3(256), 0, st, d_scalars, digits, pl, bad);
  }
  size_t lds = (size_t)pl.nb * 4;
  {  // opt in to large dynamic LDS once per process and device (c <= 16: at most 128 KB)
    static bool attr_done[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_done[dev]) {
      const int max_lds = (1 << 15) * 4;
      e = hipFuncSetAttribute((const void*)k_msm_hist, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
      if (e != hipSuccess) return e;
      e = hipFuncSetAttribute((const void*)k_msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
      if (e != hipSuccess) return e;
      e = hipFuncSetAttribute((const void*)k_msm_scan, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds + max_lds / 32 + 64);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) attr_done[dev] = true;
    }
  }
  const dim3 sort_grid = pl.xcd_map ? dim3((unsigned)(pl.Q * ((pl.nwin + 7) & ~7))) : dim3(pl.Q, pl.nwin);
  uint32_t* shared_start = (uint32_t*)(base + L.shared_start);
  Sort2 s2;
  if (small_sort) {
    static bool attr_done[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_done[dev]) {
      e = hipFuncSetAttribute((const void*)k_msm_sort_small, hipFuncAttributeMaxDynamicSharedMemorySize, MSM_SMALL_NB * 4 + MSM_SMALL_N * 2);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) attr_done[dev] = true;
    }
    hipLaunchKernelGGL(k_msm_sort_small, dim3(pl.nwin), dim3(1024), (size_t)pl.nb * 4 + (size_t)((pl.n + 1) & ~1) * 2, st, d_scalars, bstart, sorted, bad,
                       buckets, XW, (uint32_t*)(base + L.long_runs), pl);
  } else if (msm_sort2_ok(pl, std::max(pl.n, pl.n_layout), &s2)) {   // two-level sort (3b): same bucket_start / sorted
    uint32_t* ccount = counts;
    const int R = 1 << s2.lgr;
    uint32_t* region_start = ccount + (size_t)pl.nwin * pl.Q * R;
    uint32_t* oversize = region_start + (((size_t)pl.nwin * (R + 1) + 3) & ~(size_t)3);   // count, then <= nwin * R entries
    uint32_t* fcount = oversize + (((size_t)pl.nwin * R + 1 + 3) & ~(size_t)3);
    uint32_t* tmp = (uint32_t*)(base + L.sort_tmp);
    hipLaunchKernelGGL(k_sort2_count, sort_grid, dim3(1024), 0, st, digits, ccount, pl, s2);
    hipLaunchKernelGGL(k_sort2_scan, dim3(pl.nwin), dim3(R), 0, st, ccount, region_start, oversize, pl, s2);
    hipLaunchKernelGGL(k_sort2_scatter, sort_grid, dim3(1024), 0, st, digits, ccount, region_start, tmp, pl, s2);
    {
      static bool attr_done[16] = {};
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (dev < 0 || dev >= 16 || !attr_done[dev]) {
        e = hipFuncSetAttribute((const void*)k_sort2_fine_staged, hipFuncAttributeMaxDynamicSharedMemorySize, SORT2_STAGE * 4);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) attr_done[dev] = true;
      }
    }
    hipLaunchKernelGGL(k_sort2_fine_staged, dim3(R, pl.nwin), dim3(1024), (size_t)SORT2_STAGE * 4, st, tmp, region_start, oversize, bstart,
                       sorted, pl, s2);
    // regions too large for the staged kernel (skewed scalars): a small grid over the work list, empty as a rule
    hipLaunchKernelGGL(k_sort2_fine_count, dim3(SORT2_FALLBACK_BLOCKS), dim3(512), 0, st, tmp, region_start, oversize, fcount, pl, s2);
    hipLaunchKernelGGL(k_sort2_fine_place, dim3(SORT2_FALLBACK_BLOCKS), dim3(512), 0, st, tmp, region_start, oversize, fcount, bstart, sorted, pl, s2);
  } else {
    hipLaunchKernelGGL(k_msm_hist, sort_grid, dim3(1024), lds, st, digits, counts, pl);
    {
      static const int split_knob = knob("NCG_MSM_TOTALS_SPLIT", -1);   // A/B builds: force on (1) / off (0)
      if (split_knob >= 0 ? split_knob != 0 : pl.Q >= 64)
        hipLaunchKernelGGL(k_msm_bucket_totals_split, dim3((pl.nb + 31) / 32, pl.nwin), dim3(256), 0, st, counts, bstart, pl);
      else
        hipLaunchKernelGGL(k_msm_bucket_totals, dim3((pl.nb + 255) / 256, pl.nwin), dim3(256), 0, st, counts, bstart, pl);
    }
    if (pl.shared) {
      hipLaunchKernelGGL(k_msm_shared_totals, dim3((pl.nb + 255) / 256), dim3(256), 0, st, bstart, shared_start, pl);
      hipLaunchKernelGGL(k_msm_scan, dim3(1), dim3(1024), (size_t)(pl.nb + pl.nb / 32 + 1) * 4, st, shared_start, pl);  // window 0 of a 1-window array
      hipLaunchKernelGGL(k_msm_shared_starts, dim3((pl.nb + 255) / 256), dim3(256), 0, st, bstart, shared_start, pl);
    } else {
      hipLaunchKernelGGL(k_msm_scan, dim3(pl.nwin), dim3(1024), (size_t)(pl.nb + pl.nb / 32 + 1) * 4, st, bstart, pl);
    }
    {
      const int passes = std::max(1, std::min(pl.scatter_passes, pl.nb / 256));
      const int per = (pl.nb + passes - 1) / passes;
      for (int p = 0; p < passes; p++) {
        const int b_lo = p * per, b_hi = std::min(pl.nb, b_lo + per);
        if (b_lo >= b_hi) break;
        hipLaunchKernelGGL(k_msm_scatter, sort_grid, dim3(1024), (size_t)(b_hi - b_lo) * 4, st, digits, counts, bstart, sorted, pl, b_lo, b_hi);
      }
    }
  }
  // from here on: the accumulate view (shared-bucket mode: ONE window of nwin * n entries whose starts are shared_start)
  const MsmPlan av = msm_acc_view(pl);
  const uint32_t* acc_start = pl.shared ? shared_start : bstart;
  {
    MsmSeg sg = msm_seg(av);
    uint32_t* part_pts = (uint32_t*)(base + L.part_pts);
    int* part_meta = (int*)(base + L.part_meta);
    if (part_first && !small_sort) {
      e = hipMemsetAsync(buckets, 0, (size_t)av.nwin * av.nb * XW * 4, st);  // empty buckets = infinity
      if (e != hipSuccess) return e;
    }
    dim3 grid((unsigned)((((size_t)sg.nseg << LS) + 255) / 256), av.nwin);
    if (forked) {
      e = hipStreamWaitEvent(st, side->join, 0);
      if (e != hipSuccess) return e;
      side_guard.armed = false;
    }
    if (side && side->pts_ready) {  // host-pointer path: the points were still crossing PCIe while the sort ran
      e = hipStreamWaitEvent(st, side->pts_ready, 0);
      if (e != hipSuccess) return e;
    }
    {
      // A grid of at most two workgroups per CU is latency-bound (every lane's chain of `seg` additions on a wave that has its SIMD
      // to itself, or shares it with one other), and the dispatcher does not spread workgroups evenly: 238 workgroups on 256 CUs ran
      // 18.5 us per addition where 104 ran 12.  An LDS reservation the kernel never touches makes the placement explicit: with
      // 96 KB per workgroup a CU (160 KB) takes one, with 56 KB two.
      static const int spread_knob = knob("NCG_MSM_ACCUM_SPREAD", 1);   // A/B builds: 0 = off
      const size_t wgs = (size_t)grid.x * grid.y;
      size_t reserve = 0;
      if (spread_knob && wgs <= 256) reserve = 96 * 1024;
      else if (spread_knob && wgs <= 512) reserve = 56 * 1024;
      if (reserve) {
        static bool attr_done[16] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 16 || !attr_done[dev]) {
          e = hipFuncSetAttribute((const void*)k_msm_accum<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
          if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_msm_accum<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
          if (e != hipSuccess) return e;
          if (dev >= 0 && dev < 16) attr_done[dev] = true;
        }
      }
      const int top_local = (pl.nwin_total ? pl.nwin_total : pl.nwin) - 1 - pl.w0;
      if (pl.top_tb && top_local >= 0 && top_local < pl.nwin)   // this launch holds a spread top window: it may be sparse
        hipLaunchKernelGGL((k_msm_accum<D, true>), grid, dim3(256), reserve, st, pts_mont, sorted, acc_start, buckets, part_pts, part_meta, av, sg);
      else
        hipLaunchKernelGGL((k_msm_accum<D, false>), grid, dim3(256), reserve, st, pts_mont, sorted, acc_start, buckets, part_pts, part_meta, av, sg);
    }
    uint32_t* long_runs = (uint32_t*)(base + L.long_runs);
    if (!small_sort) {
      e = hipMemsetAsync(long_runs, 0, 16, st);
      if (e != hipSuccess) return e;
    }
    // How many heads the owner of a run adds itself.  A bucket of m entries cut by lanes of `seg` entries has up to
    // ceil(m / seg) heads; m is n / nb on average and rarely above m + 4 sqrt(m).  The seg that fills the chip in one
    // round can be a third of that (verified G1 set of 2^17 points: 64 entries per bucket, seg 20), and with a fixed
    // limit of 2 EVERY bucket went to the work list - 40 000 runs through 512 workgroups, 2.9 ms against 1.0 ms for the
    // generic path.  A serial head costs one dependent addition (3 us); the work list is for real outliers (equal
    // scalars, a short top window).
    const double m_avg = (double)av.n / (double)av.nb;
    const int heads_typ = (int)std::ceil((m_avg + 4.0 * std::sqrt(m_avg) + 1.0) / (double)sg.seg);
    const int run_serial_auto = std::max(knob("NCG_MSM_RUN_SERIAL", MSM_RUN_SERIAL), std::min(8, heads_typ));
    const bool rs_forced = pl.run_serial_override >= 0;  // ncg_msm_set_tuning: exactly this many (tests force the work list)
    const int run_serial = rs_forced ? pl.run_serial_override : run_serial_auto;
    // the threshold the fix-up kernel actually runs with, decided ONCE (the trace reported another value than the launch
    // used for per-window plans on the cooperative-units kernel)
    static const int units_knob = knob("NCG_MSM_MERGE_UNITS", -1);   // A/B builds: force on (1) / off (0)
    const bool merge_units = (pl.shared || (units_knob >= 0 ? units_knob != 0 : run_serial_auto >= 3)) && CoopOK<D>::value;
    static const int tree_knob = knob("NCG_MSM_MERGE_TREE", 1);   // A/B builds: 0 = one unit per bucket
    // short segments (plans that do not fill the chip): a tree of MERGE_TREE_U units per bucket, up to 16 pieces
    const bool merge_tree = merge_units && tree_knob && sg.seg < 16;
    const int run_serial_eff = rs_forced ? run_serial : std::max(run_serial, merge_tree ? 15 : (merge_units || pl.shared) ? MSM_RUN_SERIAL_SHARED : MSM_RUN_SERIAL);
    if (pl.trace) {
      MsmTrace& tr = *pl.trace;
      tr.c = pl.c; tr.nwin = pl.nwin; tr.nb = pl.nb; tr.seg = sg.seg; tr.nseg = sg.nseg; tr.w0 = pl.w0;
      tr.nwin_total = pl.nwin_total ? pl.nwin_total : pl.nwin;
      tr.run_serial = run_serial_eff;
      tr.d_long_runs = (const uint32_t*)(base + L.long_runs);
    }
    // shared-bucket mode: every bucket holds nwin * n / nb entries, i.e. a handful of pieces - all of them, so their
    // owners add them serially (fully parallel over the buckets), as cooperative groups where the curve has them;
    // the work list is for the outliers only
    // ... and the same kernel for per-window plans whose buckets are cut into three or more pieces as a rule (lanes of `seg`
    // entries against buckets of n / nb: the two-window ranks of a window-sharded MSM run seg = 16 against 32-entry buckets):
    // the owner's serial chain of 3-4 single-lane additions (~30 us each on a lone wave) was the longest kernel of such a share
    // after the accumulate itself (138 us of 0.88 ms); four lanes per addition shorten every link of it
    if (merge_tree) {
      constexpr bool MCOOP = CoopOK<D>::value;
      using K = TailOps<D, MCOOP>;
      const dim3 mgrid((unsigned)((((size_t)av.nb << (K::UNIT_SHIFT + MERGE_TREE_ULOG)) + 255) / 256), av.nwin);
      hipLaunchKernelGGL((k_msm_fixup_merge_tree<D, MCOOP>), mgrid, dim3(256), (size_t)(256 >> K::UNIT_SHIFT) * K::LDS_WORDS * 4, st, part_pts,
                         part_meta, acc_start, buckets, av, sg, long_runs, run_serial_eff);
    } else if (merge_units) {
      constexpr bool MCOOP = CoopOK<D>::value;
      using K = TailOps<D, MCOOP>;
      const dim3 mgrid((unsigned)((((size_t)av.nb << K::UNIT_SHIFT) + 255) / 256), av.nwin);
      hipLaunchKernelGGL((k_msm_fixup_merge_units<D, MCOOP>), mgrid, dim3(256), (size_t)(256 >> K::UNIT_SHIFT) * K::LDS_WORDS * 4, st, part_pts,
                         part_meta, acc_start, buckets, av, sg, long_runs,
                         // up to 8 pieces per bucket stay with the bucket's own unit (a cooperative addition is ~8 us; the work list costs
                         // a launch-wide 120 us as soon as many buckets overflow - the top window of 254-bit scalars holds 64-entry buckets)
                         run_serial_eff);
    } else {
      hipLaunchKernelGGL(k_msm_fixup_merge<D>, grid, dim3(256), 0, st, part_pts, part_meta, acc_start, buckets, av, sg, long_runs, run_serial_eff);
    }
    {
      constexpr bool LCOOP = CoopOK<D>::value;
      using K = TailOps<D, LCOOP>;
      const size_t lds_b = (size_t)(MSM_TAIL_THREADS >> K::UNIT_SHIFT) * K::LDS_WORDS * 4;
      static bool attr_done[16] = {};
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (dev < 0 || dev >= 16 || !attr_done[dev]) {
        e = hipFuncSetAttribute((const void*)k_msm_fixup_long<D, LCOOP>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) attr_done[dev] = true;
      }
      hipLaunchKernelGGL((k_msm_fixup_long<D, LCOOP>), dim3(MSM_LONG_BLOCKS), dim3(MSM_TAIL_THREADS), lds_b, st, part_pts, buckets, av, sg,
                         long_runs);
    }
  }
  if (!part_last) {  // more parts follow: the buckets stay as they are
    *d_fin = nullptr;
    if (d_bad) *d_bad = bad;
    return hipGetLastError();
  }
  // fold: nb -> 1 per window in c-1 levels.  Wide levels: one item per addition (throughput); levels that no longer
  // fill the chip: four items per addition (latency, msm_coop.hpp); the last levels + the grouping: k_msm_tail.
  // cooperative form wherever the group offers it (compile-time); NCG_MSM_COOP_LEVEL=0 keeps the separate level
  // launches on the one-item-per-addition kernel (A/B of the level kernel only)
  constexpr bool CAN_COOP = CoopOK<D>::value;
  constexpr bool coop = CAN_COOP;
  const int tail_units = MSM_TAIL_THREADS >> (LS + (coop ? 2 : 0));
  // lanes the chip keeps resident for these kernels (2 waves/SIMD): beyond that the cooperative form costs throughput
  const long coop_max_tasks = (65536L * 2) >> (LS + 2);
  const uint32_t* cur = buckets;
  int narr = 1, n_in = av.nb, flip = 0;
  {
    static bool attr_done[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_done[dev]) {
      const int max_lds = 128 * 1024;
      e = hipFuncSetAttribute((const void*)k_msm_tail<D, CAN_COOP>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) attr_done[dev] = true;
    }
  }
  // a level that needs more than `tail_rounds` rounds of the tail workgroup's units is a launch of its own: chip-wide it costs
  // 9-13 us, inside the tail 8-12 us PER ROUND (one workgroup per window)
  static const int tail_rounds = std::max(1, knob("NCG_MSM_TAIL_ROUNDS", NCG_TAIL_ROUNDS));
  while (n_in > 1 && (long)(narr + 1) * (n_in >> 1) > (long)tail_rounds * tail_units) {
    const long total = (long)(narr + 1) * av.nwin * (n_in >> 1);
    bool done = false;
    if constexpr (CAN_COOP) {
      const bool coop_level = knob("NCG_MSM_COOP_LEVEL", 1) != 0;
      if (coop && coop_level && total <= coop_max_tasks) {
        using K = TailOps<D, true>;
        const size_t lds_b = (size_t)(256 >> K::UNIT_SHIFT) * COOP_SLOTS * G::FW * 4;
        hipLaunchKernelGGL(k_msm_reduce_level_coop<D>, dim3((unsigned)(((total << K::UNIT_SHIFT) + 255) / 256)), dim3(256), lds_b, st,
                           cur, red[flip], narr, av.nwin, n_in);
        done = true;
      }
    }
    if (!done)
      hipLaunchKernelGGL(k_msm_reduce_level<D>, dim3((unsigned)(((total << LS) + 255) / 256)), dim3(256), 0, st, cur, red[flip],
                         narr, av.nwin, n_in);
    cur = red[flip];
    flip ^= 1;
    narr++;
    n_in >>= 1;
  }
  const int ng = msm_ngroups(av.c);
  {
    uint32_t* t0 = (uint32_t*)(base + L.tail0);
    uint32_t* t1 = (uint32_t*)(base + L.tail1);
    uint32_t* fin = (uint32_t*)(base + L.fin);
    using K = TailOps<D, CAN_COOP>;
    hipLaunchKernelGGL((k_msm_tail<D, CAN_COOP>), dim3(av.nwin), dim3(MSM_TAIL_THREADS), (size_t)tail_units * K::LDS_WORDS * 4, st, cur,
                       t0, t1, fin, narr, av.nwin, n_in, MSM_GROUP, ng,
                       pl.top_tb ? (pl.nwin_total ? pl.nwin_total : pl.nwin) - 1 - pl.w0 : -1, pl.top_tb, pl.tail_flag, pl.tail_gen);
    cur = fin;
  }
  *d_fin = cur;
  if (d_bad) *d_bad = bad;
  return hipGetLastError();
}

template <class C>
static size_t msm_fin_words_t(const MsmPlan& pl) {
  return (size_t)msm_ngroups(pl.c) * msm_acc_view(pl).nwin * MsmGroup<C>::ACC_WORDS;
}

template <class C> struct FinishHelpers { static constexpr bool value = false; };   // curves whose host finish runs over helper threads
template <> struct FinishHelpers<CurveG1> { static constexpr bool value = true; };
template <> struct FinishHelpers<CurveG2> { static constexpr bool value = true; };
// Finish: the grouped window sums come to the host (one small copy), Horner, canonical affine output.
// Synchronises `st`.
template <class C>
static hipError_t msm_finish_t(const MsmPlan& pl, const uint32_t* cur, uint32_t* out_affine_host, uint8_t* out_inf_host,
                               hipStream_t st, const uint32_t* d_bad = nullptr, uint32_t* bad_host = nullptr) {
  using G = MsmGroup<C>;
  constexpr int XW = G::ACC_WORDS;
  const int ng = msm_ngroups(pl.c);
  const int fin_nwin = msm_acc_view(pl).nwin;  // shared-bucket mode: one window, no Horner across windows
  hipError_t e;
  // the surviving points land in a small pinned buffer (one per thread, reused): a pageable target
  // would go through the runtime's staging copy
  const size_t fin_words = (size_t)ng * fin_nwin * XW;
  static thread_local uint32_t* pinned = nullptr;
  static thread_local size_t pinned_words = 0;
  if (pinned_words < fin_words) {
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
    pinned_words = 0;
    if (hipHostMalloc((void**)&pinned, fin_words * 4, hipHostMallocPortable) == hipSuccess) pinned_words = fin_words;
    else (void)hipGetLastError();
  }
  std::vector<uint32_t> fin(fin_words);
  uint32_t* land = pinned_words >= fin_words ? pinned : fin.data();
  e = hipMemcpyAsync(land, cur, fin_words * 4, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  if (d_bad && bad_host) {
    e = hipMemcpyAsync(bad_host, d_bad, 4, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
  }
  if (pl.tail_flag && FinishHelpers<C>::value && msm_host64_enabled() && msm_finish_threads_enabled() && h64::finish_threads_override() == 1) {
    // the helper threads of the host finish take 10-50 us to come out of their sleep: wake them when the tail kernel STARTS (it
    // runs ~0.1 ms), not when the stream is done.  Polling a word of pinned memory costs what hipStreamSynchronize's own spin costs.
    volatile uint32_t* flag = pl.tail_flag;
    for (unsigned it = 0; *flag != pl.tail_gen; it++) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      if ((it & 127u) == 127u && hipStreamQuery(st) != hipErrorNotReady) break;
    }
    h64::FinishPool::get().wake();
  }
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  if (land != fin.data()) std::copy(land, land + fin_words, fin.begin());
  static const bool timing = knob_set("NCG_TIMING");
  auto t0 = std::chrono::steady_clock::now();
  msm_host_finish_any<C>(fin.data(), pl.c, fin_nwin, out_affine_host, out_inf_host);
  if (timing) {
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[ncg] msm host finish: %.1f us (c=%d nwin=%d)\n",
            std::chrono::duration<double, std::micro>(t1 - t0).count(), pl.c, fin_nwin);
  }
  return hipSuccess;
}

// Asynchronous form (several MSMs in flight, api.hip `ncg_msm_async_*`): the device phase, then the grouped sums and the
// scalar-range flag travel to `land` (pinned host memory: fin words, then one flag word) on the same stream.  Nothing is
// synchronised here; the caller waits on its own event and runs msm_finish_host on `land`.
template <class C>
static hipError_t msm_enqueue_t(const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws, uint32_t* land,
                                hipStream_t st, const MsmSide* side) {
  const uint32_t *d_fin = nullptr, *d_bad = nullptr;
  hipError_t e = msm_device_t<C>(pl, d_pts, d_scalars, ws, &d_fin, st, &d_bad, side);
  if (e != hipSuccess) return e;
  const size_t fin_words = msm_fin_words_t<C>(pl);
  e = hipMemcpyAsync(land, d_fin, fin_words * 4, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  return hipMemcpyAsync(land + fin_words, d_bad, 4, hipMemcpyDeviceToHost, st);
}

template <class C>
static hipError_t msm_run_t(const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws,
                            uint32_t* out_affine_host, uint8_t* out_inf_host, hipStream_t st, uint32_t* bad_index,
                            const MsmSide* side) {
  const uint32_t *d_fin = nullptr, *d_bad = nullptr;
  MsmPlan plr = pl;
  if (FinishHelpers<C>::value && msm_acc_view(pl).nwin >= 4) {   // one pinned word per calling thread: "the tail kernel has started"
    static thread_local uint32_t* flag = nullptr;
    static thread_local uint32_t gen = 0;
    if (!flag) {
      if (hipHostMalloc((void**)&flag, 64, hipHostMallocPortable) == hipSuccess) *flag = 0;
      else { flag = nullptr; (void)hipGetLastError(); }
    }
    if (flag) {
      plr.tail_flag = flag;
      plr.tail_gen = ++gen ? gen : ++gen;   // never 0
    }
  }
  hipError_t e = msm_device_t<C>(plr, d_pts, d_scalars, ws, &d_fin, st, &d_bad, side);
  if (e != hipSuccess) return e;
  return msm_finish_t<C>(plr, d_fin, out_affine_host, out_inf_host, st, d_bad, bad_index);
}

template <class C>
static hipError_t msm_sum_partials_t(uint32_t* d_gathered, int nparts, size_t npoints, uint32_t* d_out, hipStream_t st) {
  using D = typename DeviceCurve<C>::type;
  constexpr bool COOP = CoopOK<D>::value;
  using T = TailOps<D, COOP>;
  hipLaunchKernelGGL((k_msm_sum_partials<D, COOP>), dim3((unsigned)npoints), dim3(256), (size_t)(256 >> T::UNIT_SHIFT) * T::SCRATCH_WORDS * 4,
                     st, d_gathered, d_out, nparts, (int)npoints);
  return hipGetLastError();
}

template <class C>
static hipError_t msm_points_to_stored_t(const uint32_t* d_pts_wire, int n, uint32_t* d_out, hipStream_t st) {
  using D = typename DeviceCurve<C>::type;
  constexpr int LS = LaneShift<D>::value;
  hipLaunchKernelGGL(k_points_to_mont<D>, dim3((unsigned)((((size_t)n << LS) + 255) / 256)), dim3(256), 0, st, d_pts_wire, d_out, n);
  return hipGetLastError();
}
// ---- bls12-381 G1 lives in msm_g1.o (see the head of this file): the three templates that launch curve kernels forward there
hipError_t msm_g1_device(const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws, const uint32_t** d_fin,
                         hipStream_t st, const uint32_t** d_bad, const MsmSide* side);
hipError_t msm_g1_sum_partials(uint32_t* d_gathered, int nparts, size_t npoints, uint32_t* d_out, hipStream_t st);
hipError_t msm_g1_points_to_stored(const uint32_t* d_pts_wire, int n, uint32_t* d_out, hipStream_t st);
#ifdef NCG_MSM_TU_G1
hipError_t msm_g1_device(const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws, const uint32_t** d_fin,
                         hipStream_t st, const uint32_t** d_bad, const MsmSide* side) {
  return msm_device_t<CurveG1>(pl, d_pts, d_scalars, ws, d_fin, st, d_bad, side);
}
hipError_t msm_g1_sum_partials(uint32_t* d_gathered, int nparts, size_t npoints, uint32_t* d_out, hipStream_t st) {
  return msm_sum_partials_t<CurveG1>(d_gathered, nparts, npoints, d_out, st);
}
hipError_t msm_g1_points_to_stored(const uint32_t* d_pts_wire, int n, uint32_t* d_out, hipStream_t st) {
  return msm_points_to_stored_t<CurveG1>(d_pts_wire, n, d_out, st);
}
#else
template <>
hipError_t msm_device_t<CurveG1>(const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws, const uint32_t** d_fin,
                                 hipStream_t st, const uint32_t** d_bad, const MsmSide* side) {
  return msm_g1_device(pl, d_pts, d_scalars, ws, d_fin, st, d_bad, side);
}
template <>
hipError_t msm_sum_partials_t<CurveG1>(uint32_t* d_gathered, int nparts, size_t npoints, uint32_t* d_out, hipStream_t st) {
  return msm_g1_sum_partials(d_gathered, nparts, npoints, d_out, st);
}
template <>
hipError_t msm_points_to_stored_t<CurveG1>(const uint32_t* d_pts_wire, int n, uint32_t* d_out, hipStream_t st) {
  return msm_g1_points_to_stored(d_pts_wire, n, d_out, st);
}

#define NCG_MSM_DISPATCH(curve, CALL)                       \
  switch (curve) {                                          \
    case CURVE_SECP256K1: return CALL(CurveSecp);           \
    case CURVE_BLS12_381_G1: return CALL(CurveG1);          \
    case CURVE_BLS12_381_G2: return CALL(CurveG2);          \
    case CURVE_ED25519: return CALL(CurveEd);               \
    default: return hipErrorInvalidValue;                   \
  }

hipError_t msm_device_phase(int curve, const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws,
                            const uint32_t** d_fin, hipStream_t st, const uint32_t** d_bad, const MsmSide* side) {
#define CALL(C) msm_device_t<C>(pl, d_pts, d_scalars, ws, d_fin, st, d_bad, side)
  NCG_MSM_DISPATCH(curve, CALL)
#undef CALL
}
hipError_t msm_finish(int curve, const MsmPlan& pl, const uint32_t* d_fin, uint32_t* out_affine_host,
                      uint8_t* out_inf_host, hipStream_t st, const uint32_t* d_bad, uint32_t* bad_host) {
#define CALL(C) msm_finish_t<C>(pl, d_fin, out_affine_host, out_inf_host, st, d_bad, bad_host)
  NCG_MSM_DISPATCH(curve, CALL)
#undef CALL
}
hipError_t msm_sum_partials(int curve, uint32_t* d_gathered, int nparts, size_t npoints, uint32_t* d_out,
                            hipStream_t st) {
#define CALL(C) msm_sum_partials_t<C>(d_gathered, nparts, npoints, d_out, st)
  NCG_MSM_DISPATCH(curve, CALL)
#undef CALL
}
hipError_t msm_enqueue(int curve, const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws, uint32_t* land,
                       hipStream_t st, const MsmSide* side) {
#define CALL(C) msm_enqueue_t<C>(pl, d_pts, d_scalars, ws, land, st, side)
  NCG_MSM_DISPATCH(curve, CALL)
#undef CALL
}
// Horner over [ngroups(c)][nwin] grouped window sums in HOST memory (device storage format) + canonical affine output
void msm_finish_prewake(int curve) {
  if ((curve == CURVE_BLS12_381_G1 || curve == CURVE_BLS12_381_G2) && msm_host64_enabled() && msm_finish_threads_enabled() &&
      h64::finish_threads_override() == 1)
    h64::FinishPool::get().wake();
}
void msm_finish_host(int curve, int c, int nwin, const uint32_t* fin_host, uint32_t* out_affine_host, uint8_t* out_inf_host) {
  switch (curve) {
    case CURVE_SECP256K1: return msm_host_finish_any<CurveSecp>(fin_host, c, nwin, out_affine_host, out_inf_host);
    case CURVE_BLS12_381_G1: return msm_host_finish_any<CurveG1>(fin_host, c, nwin, out_affine_host, out_inf_host);
    case CURVE_BLS12_381_G2: return msm_host_finish_any<CurveG2>(fin_host, c, nwin, out_affine_host, out_inf_host);
    case CURVE_ED25519: return msm_host_finish_any<CurveEd>(fin_host, c, nwin, out_affine_host, out_inf_host);
    default: return;
  }
}
size_t msm_fin_words(int curve, const MsmPlan& pl) {
  switch (curve) {
    case CURVE_SECP256K1: return msm_fin_words_t<CurveSecp>(pl);
    case CURVE_BLS12_381_G1: return msm_fin_words_t<CurveG1>(pl);
    case CURVE_BLS12_381_G2: return msm_fin_words_t<CurveG2>(pl);
    case CURVE_ED25519: return msm_fin_words_t<CurveEd>(pl);
    default: return 0;
  }
}
size_t msm_acc_words(int curve) {
  switch (curve) {
    case CURVE_SECP256K1: return MsmGroup<CurveSecp>::ACC_WORDS;
    case CURVE_BLS12_381_G1: return MsmGroup<CurveG1>::ACC_WORDS;
    case CURVE_BLS12_381_G2: return MsmGroup<CurveG2>::ACC_WORDS;
    case CURVE_ED25519: return MsmGroup<CurveEd>::ACC_WORDS;
    default: return 0;
  }
}

size_t msm_workspace_bytes(int curve, const MsmPlan& pl) {
  switch (curve) {
    case CURVE_SECP256K1: return msm_layout<CurveSecp>(pl).total;
    case CURVE_BLS12_381_G1: return msm_layout<CurveG1>(pl).total;
    case CURVE_BLS12_381_G2: return msm_layout<CurveG2>(pl).total;
    case CURVE_ED25519: return msm_layout<CurveEd>(pl).total;
    default: return 0;
  }
}

size_t msm_stored_words_per_point(int curve) {
  switch (curve) {
    case CURVE_SECP256K1: return MsmGroup<CurveSecp>::AFF_WORDS;
    case CURVE_BLS12_381_G1: return MsmGroup<CurveG1>::AFF_WORDS;
    case CURVE_BLS12_381_G2: return MsmGroup<CurveG2>::AFF_WORDS;
    case CURVE_ED25519: return MsmGroup<CurveEd>::AFF_WORDS;
    default: return 0;
  }
}
hipError_t msm_points_to_stored(int curve, const uint32_t* d_pts_wire, int n, uint32_t* d_out, hipStream_t st) {
#define CALL(C) msm_points_to_stored_t<C>(d_pts_wire, n, d_out, st)
  NCG_MSM_DISPATCH(curve, CALL)
#undef CALL
}

hipError_t msm_run(int curve, const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws,
                   uint32_t* out_affine_host, uint8_t* out_inf_host, hipStream_t st, uint32_t* bad_index,
                   const MsmSide* side) {
  uint32_t dummy = 0xFFFFFFFFu;
  if (!bad_index) bad_index = &dummy;
  *bad_index = 0xFFFFFFFFu;
  switch (curve) {
    case CURVE_SECP256K1: return msm_run_t<CurveSecp>(pl, d_pts, d_scalars, ws, out_affine_host, out_inf_host, st, bad_index, side);
    case CURVE_BLS12_381_G1: return msm_run_t<CurveG1>(pl, d_pts, d_scalars, ws, out_affine_host, out_inf_host, st, bad_index, side);
    case CURVE_BLS12_381_G2: return msm_run_t<CurveG2>(pl, d_pts, d_scalars, ws, out_affine_host, out_inf_host, st, bad_index, side);
    case CURVE_ED25519: return msm_run_t<CurveEd>(pl, d_pts, d_scalars, ws, out_affine_host, out_inf_host, st, bad_index, side);
    default: return hipErrorInvalidValue;
  }
}

#endif  // NCG_MSM_TU_G1

}  // namespace ncg
