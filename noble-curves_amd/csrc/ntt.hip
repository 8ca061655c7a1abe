// Radix-2 number-theoretic transform over the bls12-381 scalar field Fr.
//
// Reference: FFT(roots, Fr).direct / .inverse (src/abstract/fft.ts:518-577) over FFTCore
// (:422-480) with the tables of rootsOfUnity (:230-312).  The four input/output orderings of the
// reference are the same map up to bit-reversal permutations BR:
//   (brpInput, brpOutput) = (0,0): D(x)   (0,1): BR(D(x))   (1,0): D(BR(x))   (1,1): BR(D(BR(x)))
// with D the natural-order DFT y[k] = sum_i x[i] w^(ik) over the table in use (roots(bits) for
// direct, its reversal inverse(bits) for inverse, which also scales by 1/N, :566-573).  Field
// arithmetic is exact, so any evaluation order gives the reference's values bit for bit.
//
// Device plan: the stages are grouped into passes of up to 10 that run on an LDS tile (limb-major,
// conflict-free), one HBM round trip per pass.  Every butterfly is of the multiply-first kind
// (t = b w, a + t, a - t):
//   * bit-reversed in -> natural out: the reference's DIT loop (:470-473), stages ascending, twiddle
//     roots[j * (N >> s)] for position j inside the block;
//   * natural in -> bit-reversed out: the same butterfly with the stages DESCENDING and one twiddle per
//     block, roots[brev(block) << (s - 1)] - the evaluation-tree form of the transform (f mod (x^m - c)
//     splits into f_lo +- sqrt(c) f_hi).  It computes the map of the reference's DIF loop (:476-478)
//     - field arithmetic is exact, so the values are the same - but keeps the lazily reduced operands
//     of fr29.hpp small: the a + b of the DIF butterfly doubles the value every stage, a + t adds 3 r.
// The remaining bit reversal of the (0,0) and (1,1) forms is folded into the last pass's store.
// Arithmetic: fr29.hpp (9 x 29-bit limbs, R = 2^261, lazy sums; about 1 200 issue cycles per
// butterfly against 1 700 for the 8 x 32-bit Montgomery form this file used before round 3).  Data is
// canonical in HBM at both ends and a value below 2^256 (8 words) between passes; twiddles are kept as
// w 2^261 mod r, and mont(a, w 2^261) = a w.  The twiddle table is the reference's natural order
// roots(bits) (N entries); inverse(bits)[k] = roots[(N - k) mod N] is index arithmetic.
#include <vector>

#include <algorithm>
#include "knobs.hpp"
#include "fp.hpp"
#include "fr29.hpp"
#include "host_api.hpp"

namespace ncg {

using Fr = Fp<ParamsBlsR>;

struct NttPass {
  int n;          // log2 N
  int s_lo;       // lowest stage of this pass (stage s has sub-transform length m = 2^s)
  int T;          // number of stages in the pass
  int logC;       // log2 of the contiguous run kept per tile row (0 when s_lo == 1)
  int dit;        // 1: DIT butterflies, stages ascending; 0: DIF, stages descending
  int inverse;    // use roots[(N - k) mod N]
  int brp_store;  // store to the bit-reversed index
  int scale;      // multiply by 1/N on store
  int canon;      // last pass: store canonical residues
  int colhi;      // the 2^logC columns of the tile are the TOP index bits (rows = the low T bits): the lowest pass
                  // of a natural -> natural transform, whose bit-reversed store then writes runs of 2^logC elements
  int tshift;     // log2(table size) - n
};

NCG_DI Fr fr_load_g(const uint32_t* __restrict__ p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
NCG_DI void fr_store_g(uint32_t* __restrict__ p, const Fr& r) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
  q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// One pass: tile = 2^T rows (the index bits this pass transforms) x 2^logC contiguous elements.
// src/dst hold `batch` polynomials of N elements (8 LE words each); blockIdx.y = polynomial.
// The three phases are functions of (tid, nthreads, block) so that the host twin (ntt_pass_host, used by
// hosttest.hip) executes the same code serially.
struct NttTile {
  int C, L;
  uint32_t lochunk, hi;
  size_t N, poly, base;
};
NCG_DI NttTile ntt_tile(const NttPass& ps, uint32_t bx, uint32_t by) {
  NttTile t;
  t.C = 1 << ps.logC;
  t.L = ps.s_lo - 1;  // index bits below the tile rows
  t.N = (size_t)1 << ps.n;
  t.poly = (size_t)by * t.N * 8;
  if (ps.colhi) {  // s_lo == 1: element (row, col) is index col << (n - logC) | bx << T | row
    t.lochunk = 0;
    t.hi = bx;
    t.base = (size_t)bx << ps.T;
    return t;
  }
  t.lochunk = bx & ((1u << (t.L - ps.logC)) - 1u);
  t.hi = bx >> (t.L - ps.logC);
  t.base = ((size_t)t.hi << (t.L + ps.T)) | ((size_t)t.lochunk << ps.logC);
  return t;
}
// global index of tile element e
NCG_DI size_t ntt_tile_index(const NttPass& ps, const NttTile& t, int e) {
  if (ps.colhi) return t.base | ((size_t)(e & (t.C - 1)) << (ps.n - ps.logC)) | (size_t)(e >> ps.logC);
  return t.base | ((size_t)(e >> ps.logC) << t.L) | (size_t)(e & (t.C - 1));
}
NCG_DI uint32_t ntt_brev32(uint32_t x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __brev(x);
#else
  uint32_t r = 0;
  for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i);
  return r;
#endif
}
NCG_DI Fr29 fr29_load_g(const uint32_t* __restrict__ p) {
  const Fr x = fr_load_g(p);
  return fr29_from_words(x.v);
}
// twiddle table entry: the 9 limbs of w 2^261 mod r, no conversion on load.  Nine words per entry since the third session of round 6 (36 bytes at
// 4-byte alignment; twelve words = three aligned 16-byte loads before): a 2^22 transform reads its table about twice, a quarter less of it is 1-3 % of
// the transform (profiles/r06_ntt_tw36_ab.txt) and 50 MB less device memory per 2^22 table
#ifndef NCG_NTT_TW_WORDS
#define NCG_NTT_TW_WORDS 9
#endif
constexpr int NTT_TW = NCG_NTT_TW_WORDS;   // 12: three aligned 16-byte loads per twiddle; 9: 36-byte entries at 4-byte alignment (a quarter less table traffic)
struct __attribute__((packed, aligned(4))) NttTw9 { uint32_t v[9]; };
NCG_DI Fr29 ntt_load_tw(const uint32_t* __restrict__ p) {
  Fr29 r;
  if constexpr (NTT_TW == 9) {
    const NttTw9 t = *reinterpret_cast<const NttTw9*>(p);
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = t.v[i];
    return r;
  }
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1], c = q[2];
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  r.v[8] = c.x;
  return r;
}
NCG_DI void ntt_store_tw(uint32_t* __restrict__ p, const Fr& canonical) {
  const Fr29 l = fr29_from_words(canonical.v);
  if constexpr (NTT_TW == 9) {
    NttTw9 t;
#pragma unroll
    for (int i = 0; i < 9; i++) t.v[i] = l.v[i];
    *reinterpret_cast<NttTw9*>(p) = t;
    return;
  }
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(l.v[0], l.v[1], l.v[2], l.v[3]);
  q[1] = make_uint4(l.v[4], l.v[5], l.v[6], l.v[7]);
  q[2] = make_uint4(l.v[8], 0u, 0u, 0u);
}
template <int E, class LDS>
NCG_DI Fr29 fr29_load_l(const LDS lds, int e) {
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = lds[i * E + e];
  return r;
}
template <int E, class LDS>
NCG_DI void fr29_store_l(LDS lds, int e, const Fr29& a) {
#pragma unroll
  for (int i = 0; i < 9; i++) lds[i * E + e] = a.v[i];
}

// Work assignment inside a block: thread tid runs butterfly b = tid of every stage (E / 2 threads; one
// wave for the tiles below 128 elements), and wavefront w loads / stores the elements [128 w, 128 w + 128).
// A butterfly over tile-row bit q pairs elements 2^(q + logC) apart, and butterfly b touches the elements
// obtained by inserting a bit at position q + logC into b: for q + logC <= 6 both lie in the segment of
// the wavefront that runs b.  Such stages need no workgroup barrier - LDS operations of one wavefront
// execute in order - only the stages that pair elements across segments do (ntt_stage_cross).
NCG_DI int ntt_own_elem(int tid, int k) { return ((tid >> 6) << 7) + (tid & 63) + 64 * k; }
NCG_DI int ntt_stage_q(const NttPass& ps, int st) { return ps.dit ? st : ps.T - 1 - st; }  // tile-row bit split by stage st
NCG_DI bool ntt_stage_cross(const NttPass& ps, int st) { return ntt_stage_q(ps, st) + ps.logC > 6; }
// every twiddle of the stage is 1: the first stage of either order (j = 0 resp. block 0)
NCG_DI bool ntt_stage_trivial(const NttPass& ps, int st) {
  const int s = ps.s_lo + ntt_stage_q(ps, st);  // FFTCore stage: m = 2^s, stride = N >> s  (fft.ts:456-460)
  return ps.dit ? s == 1 : s == ps.n;
}

template <int E, class LDS>
NCG_DI void ntt_pass_load(LDS lds, int tid, const uint32_t* __restrict__ src, const NttPass& ps, const NttTile& t) {
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int e = ntt_own_elem(tid, k);
    if (e < E) {
      const size_t g = ntt_tile_index(ps, t, e);
      fr29_store_l<E>(lds, e, fr29_load_g(src + t.poly + g * 8));
    }
  }
}

struct NttBf {
  int e0, e1;    // LDS elements of the butterfly
  uint32_t pos;  // table index of its twiddle
};
NCG_DI NttBf ntt_bf_index(const NttPass& ps, const NttTile& t, int b, int st) {
  const int logC = ps.logC;
  const int q = ntt_stage_q(ps, st);
  const int s = ps.s_lo + q;
  const uint32_t c = b & (t.C - 1), bm = (uint32_t)b >> logC;
  const uint32_t low = bm & ((1u << q) - 1u);
  const uint32_t mid0 = ((bm >> q) << (q + 1)) | low;
  NttBf r;
  r.e0 = (mid0 << logC) | c;
  r.e1 = r.e0 + (1 << (q + logC));
  uint32_t pos;
  if (ps.dit) {  // j = i0 mod m/2; rootPos = j * (N >> s)  (fft.ts:463-467)
    const uint32_t j = ps.colhi ? low : (low << t.L) | (t.lochunk << logC) | c;
    pos = j << (ps.n - s);
  } else {  // block = i0 >> s (n - s bits), exponent brev(block) * m/2; s == n is the trivial stage
    uint32_t blk = (t.hi << (ps.T - 1 - q)) | (bm >> q);
    if (ps.colhi) blk |= c << (ps.n - logC - s);  // the column is the top of the index
    pos = s == ps.n ? 0u : (ntt_brev32(blk) >> (32 - (ps.n - s))) << (s - 1);
  }
  pos <<= ps.tshift;
  const uint32_t tmask = (uint32_t)((t.N << ps.tshift) - 1);
  if (ps.inverse) pos = (0u - pos) & tmask;  // inverse(bits)[k] = roots[(N - k) mod N]  (fft.ts:296-304)
  r.pos = pos;
  return r;
}
// `weak`: bring the limbs back below 2^29 + 8 on the way out (the caller tracks the limb bound: +2 per
// stage, at most 5 going in)
template <int E, class LDS>
NCG_DI void ntt_bf_run(LDS lds, const NttBf& bf, bool trivial, const Fr29& tw, bool weak) {
  const Fr29 a = fr29_load_l<E>(lds, bf.e0);
  Fr29 tt = fr29_load_l<E>(lds, bf.e1);
  if (!trivial) tt = fr29_mont(tt, tw);
  Fr29 o0 = fr29_add(a, tt), o1 = fr29_sub(a, tt);  // fft.ts:470-473
  if (weak) {
    o0 = fr29_weak(o0);
    o1 = fr29_weak(o1);
  }
  fr29_store_l<E>(lds, bf.e0, o0);
  fr29_store_l<E>(lds, bf.e1, o1);
}
template <int E, class LDS>
NCG_DI void ntt_pass_store(LDS lds, int tid, uint32_t* __restrict__ dst, const uint32_t* __restrict__ tab,
                           const NttPass& ps, const NttTile& t) {
  Fr29 ninv;
  if (ps.scale) ninv = ntt_load_tw(tab + (((size_t)t.N << ps.tshift)) * NTT_TW);  // slot after the table: (1/N) 2^261
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int e = ntt_own_elem(tid, k);
    if (e < E) {
      size_t g = ntt_tile_index(ps, t, e);
      Fr29 v = fr29_load_l<E>(lds, e);
      if (ps.scale) v = fr29_mont(v, ninv);  // fft.ts:568-570; below 1.5 r
      else v = fr29_reduce256(v);            // below 1.29 * 2^255 = 1.42 r
      if (ps.canon) v = fr29_cond_sub(v);
      Fr o;
      fr29_to_words(o.v, v);
      if (ps.brp_store) g = ps.n ? (size_t)(ntt_brev32((uint32_t)g) >> (32 - ps.n)) : 0;
      fr_store_g(dst + t.poly + g * 8, o);
    }
  }
}
// limb bound bookkeeping shared by the kernel and the host twin: stage outputs are at bound + 2; they are
// normalised when that passes 5 (a sum must fit 32 bits, a product operand may be at most 6)
NCG_DI bool ntt_stage_weak(int& bound) {
  bound += 2;
  if (bound > 5) {
    bound = 1;
    return true;
  }
  return false;
}

// LOGE = log2 of the tile's element count (compile-time: the limb planes of the tile sit at immediate LDS
// offsets).  E / 2 threads (64 for the tiles below 128 elements): 36 KB tiles run as four blocks, 32 waves per CU.
constexpr int ntt_threads(int loge) { return loge >= 7 ? 1 << (loge - 1) : 64; }
#ifdef __HIP_DEVICE_COMPILE__
// LDS hand-over between two stages: inside the wavefront when neither stage pairs across segments
__device__ __forceinline__ void ntt_sync(bool block) {
  if (block) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}
#endif
template <int LOGE>
__global__ void __launch_bounds__(ntt_threads(LOGE), 8) k_ntt_pass(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                                  const uint32_t* __restrict__ tab, NttPass ps) {
#ifdef __HIP_DEVICE_COMPILE__
  extern __shared__ uint32_t lds[];
  constexpr int E = 1 << LOGE;
  const int tid = (int)threadIdx.x;
  const bool active = tid < E / 2;
  const NttTile t = ntt_tile(ps, blockIdx.x, blockIdx.y);
  ntt_pass_load<E>(lds, tid, src, ps, t);
  ntt_sync(ntt_stage_cross(ps, 0));
  int bound = 1;
  for (int st = 0; st < ps.T; st++) {
    const bool weak = ntt_stage_weak(bound);
    const bool trivial = ntt_stage_trivial(ps, st);
    if (active) {
      const NttBf bf = ntt_bf_index(ps, t, tid, st);
      Fr29 tw;
      if (!trivial) tw = ntt_load_tw(tab + (size_t)bf.pos * NTT_TW);
      ntt_bf_run<E>(lds, bf, trivial, tw, weak);
    }
    ntt_sync(ntt_stage_cross(ps, st) || (st + 1 < ps.T && ntt_stage_cross(ps, st + 1)));
  }
  ntt_pass_store<E>(lds, tid, dst, tab, ps, t);
#endif
}

// ---- twiddle table: tab[k] = omega^k 2^261 mod r, k < N; tab[N] = (1/N) 2^261 mod r (canonical residues as 9 limbs in
// NTT_TW words: the second operand of fr29_mont).  Built with the 8 x 32-bit Montgomery arithmetic of fp.hpp (R = 2^256): a value x R
// times the plain integer K261 = 2^261 mod r is x 2^261.
constexpr int NTT_SPLIT = 12;
// small[0 .. 2^lo) = omega^j ; small[2^lo .. 2^lo + 2^hi) = omega^(j 2^lo)
__global__ void k_ntt_small_tables(const uint32_t* __restrict__ omega_wire, uint32_t* __restrict__ small, int lo, int hi) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nlo = 1 << lo, nhi = 1 << hi;
  if (t >= nlo + nhi) return;
  Fr w = fp_to_mont<ParamsBlsR>(fr_load_g(omega_wire));
  uint32_t ex = t < nlo ? (uint32_t)t : (uint32_t)(t - nlo);
  if (t >= nlo) w = fp_sqr_n<ParamsBlsR>(w, lo);
  Fr r = Fr::one();
  while (ex) {
    if (ex & 1u) r = r * w;
    w = fp_sqr<ParamsBlsR>(w);
    ex >>= 1;
  }
  fr_store_g(small + (size_t)t * 8, r);
}
__global__ void __launch_bounds__(256) k_ntt_fill_table(const uint32_t* __restrict__ small, uint32_t* __restrict__ tab,
                                                        int n, int lo) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t N = (size_t)1 << n;
  if (k > N) return;
  if (k == N) {  // 1/N = (1/2)^n
    Fr h = Fr::from_const(ParamsBlsR::INV2), r = Fr::one();
    for (int i = 0; i < n; i++) r = r * h;
    ntt_store_tw(tab + k * NTT_TW, r * Fr::from_const(Fr29PR::K261));
    return;
  }
  const Fr a = fr_load_g(small + (k & (((size_t)1 << lo) - 1)) * 8);
  const Fr b = fr_load_g(small + (((size_t)1 << lo) + (k >> lo)) * 8);
  ntt_store_tw(tab + k * NTT_TW, a * b * Fr::from_const(Fr29PR::K261));
}

// host view of one table entry (for the primitive-root probe of the API layer)
int ntt_tw_words() { return NTT_TW; }
void ntt_tw_from_canonical(const uint32_t (&w)[8], uint32_t* out) {
  Fr x;
  for (int i = 0; i < 8; i++) x.v[i] = w[i];
  ntt_store_tw(out, x);
}
size_t ntt_table_bytes(int n) { return (((size_t)1 << n) + 1) * NTT_TW * 4; }

// d_omega: device copy of the primitive 2^n-th root (canonical wire); d_small: scratch of
// ntt_small_bytes(n) bytes.  ok_host[0] = 1 iff omega^(N/2) == -1 (or n == 0 and omega == 1).
size_t ntt_small_bytes(int n) {
  const int lo = n < NTT_SPLIT ? n : NTT_SPLIT, hi = n - lo;
  return (((size_t)1 << lo) + ((size_t)1 << hi)) * 32;
}
hipError_t ntt_build_table(int n, const uint32_t* d_omega, uint32_t* d_small, uint32_t* d_tab, hipStream_t st) {
  const int lo = n < NTT_SPLIT ? n : NTT_SPLIT, hi = n - lo;
  const int cnt = (1 << lo) + (1 << hi);
  hipLaunchKernelGGL(k_ntt_small_tables, dim3((cnt + 127) / 128), dim3(128), 0, st, d_omega, d_small, lo, hi);
  const size_t N1 = ((size_t)1 << n) + 1;
  hipLaunchKernelGGL(k_ntt_fill_table, dim3((unsigned)((N1 + 255) / 256)), dim3(256), 0, st, d_small, d_tab, n, lo);
  return hipGetLastError();
}

// Stage groups, lowest first: the lowest covers up to t0max (10) stages on contiguous 2^T tiles, the others up to tmax (8)
// on 2^T x 4 tiles.  The stages are spread evenly over the fewest passes that hold them (12 = 6 + 6, not 10 + 2: a
// 2-stage pass runs tiles of 16 elements on 64-thread blocks).
int ntt_plan(int n, int (&s_lo)[8], int (&T)[8], int t0max = 10, int tmax = 8) {
  if (n == 0) return 0;
  int np = 1;
  if (n > t0max) np = 1 + (n - t0max + tmax - 1) / tmax;
  int t0 = (n + np - 1) / np;  // even share, within the lowest pass's limit
  if (t0 > t0max) t0 = t0max;
  if (n - t0 > (np - 1) * tmax) t0 = n - (np - 1) * tmax;
  int cnt = 0;
  s_lo[cnt] = 1;
  T[cnt++] = t0;
  int rem = n - t0, s = t0 + 1;
  const int groups = np - 1;
  for (int g = 0; g < groups; g++) {
    int t = (rem + (groups - g) - 1) / (groups - g);
    s_lo[cnt] = s;
    T[cnt++] = t;
    s += t;
    rem -= t;
  }
  return cnt;
}

// The passes of one transform in execution order.  flags: bit 0 inverse, bit 1 brpInput, bit 2 brpOutput.
// buf[k]: 0 = src, 1 = dst, 2 = ws for pass k's input / output.
struct NttSchedule {
  int np;
  NttPass ps[8];
  int in[8], out[8];
};
NttSchedule ntt_schedule(int n, int tab_log, int flags, int t0max = 10, int tmax = 8) {
  const bool inverse = flags & 1, brp_in = flags & 2, brp_out = flags & 4;
  NttSchedule sc;
  int s_lo[8], T[8];
  const bool dit = brp_in;
  const bool fold_brp = brp_in == brp_out;  // (0,0): natural-order butterflies then BR;  (1,1): DIT then BR
  // (0,0) in several passes: the bit-reversed store falls to the lowest pass, whose contiguous tile would scatter
  // single 32-byte elements; give it 4 columns from the top of the index instead (8 + 2 bits per tile)
  const bool colhi = fold_brp && !dit && n > t0max;
  if (colhi && t0max > 8) t0max = 8;
  sc.np = ntt_plan(n, s_lo, T, t0max, tmax);
  for (int k = 0; k < sc.np; k++) {
    const int g = dit ? k : sc.np - 1 - k;  // DIT ascends the stages, the natural-input order descends
    NttPass& ps = sc.ps[k];
    ps.n = n;
    ps.s_lo = s_lo[g];
    ps.T = T[g];
    ps.colhi = colhi && s_lo[g] == 1;
    ps.logC = s_lo[g] == 1 && !ps.colhi ? 0 : 2;
    ps.dit = dit;
    ps.inverse = inverse;
    ps.brp_store = fold_brp && k == sc.np - 1;
    ps.scale = inverse && k == sc.np - 1;
    ps.canon = k == sc.np - 1;
    ps.tshift = tab_log - n;
    if (fold_brp && sc.np > 1) {
      sc.in[k] = k == 0 ? 0 : 2;
      sc.out[k] = k == sc.np - 1 ? 1 : 2;
    } else {
      sc.in[k] = k == 0 ? 0 : 1;
      sc.out[k] = 1;
    }
  }
  return sc;
}

template <int LOGE>
static void ntt_launch(dim3 grid, hipStream_t st, const uint32_t* in, uint32_t* out, const uint32_t* tab, const NttPass& ps) {
  static const int lds_pct = std::max(100, knob("NCG_NTT_LDS_PCT", 100));   // A/B builds: occupancy experiment (170 = about half the workgroups per CU)
  hipLaunchKernelGGL((k_ntt_pass<LOGE>), grid, dim3(ntt_threads(LOGE)), std::min((size_t)65536, ((size_t)36 << LOGE) * lds_pct / 100), st, in, out, tab, ps);
}

// ws: batch * N * 32 bytes (only read when the bit reversal is folded into a multi-pass transform); src may
// equal dst.
hipError_t ntt_run(int n, size_t batch, const uint32_t* src, uint32_t* dst, uint32_t* ws, const uint32_t* tab,
                   int tab_log, int flags, hipStream_t st) {
  const size_t N = (size_t)1 << n;
  if (n == 0) {  // N = 1: identity (and 1/N = 1)
    if (src != dst) return hipMemcpyAsync(dst, src, batch * 32, hipMemcpyDeviceToDevice, st);
    return hipSuccess;
  }
  const NttSchedule sc = ntt_schedule(n, tab_log, flags);
  for (int k = 0; k < sc.np; k++) {
    const NttPass& ps = sc.ps[k];
    const uint32_t* in = sc.in[k] == 0 ? src : sc.in[k] == 1 ? dst : ws;
    uint32_t* out = sc.out[k] == 1 ? dst : ws;
    const int loge = ps.T + ps.logC;
    const dim3 grid((unsigned)(N >> loge), (unsigned)batch);
    switch (loge) {
      case 1: ntt_launch<1>(grid, st, in, out, tab, ps); break;
      case 2: ntt_launch<2>(grid, st, in, out, tab, ps); break;
      case 3: ntt_launch<3>(grid, st, in, out, tab, ps); break;
      case 4: ntt_launch<4>(grid, st, in, out, tab, ps); break;
      case 5: ntt_launch<5>(grid, st, in, out, tab, ps); break;
      case 6: ntt_launch<6>(grid, st, in, out, tab, ps); break;
      case 7: ntt_launch<7>(grid, st, in, out, tab, ps); break;
      case 8: ntt_launch<8>(grid, st, in, out, tab, ps); break;
      case 9: ntt_launch<9>(grid, st, in, out, tab, ps); break;
      case 10: ntt_launch<10>(grid, st, in, out, tab, ps); break;
      default: return hipErrorInvalidValue;  // ntt_plan never asks for more
    }
  }
  return hipGetLastError();
}

template <int LOGE>
static void ntt_host_pass_t(const uint32_t* in, uint32_t* out, const uint32_t* tab, const NttPass& ps,
                            std::vector<uint32_t>& lds) {
  constexpr int E = 1 << LOGE, nt = ntt_threads(LOGE);
  lds.assign((size_t)E * 9, 0u);
  const uint32_t tiles = (uint32_t)(((size_t)1 << ps.n) >> LOGE);
  for (uint32_t bx = 0; bx < tiles; bx++) {
    const NttTile t = ntt_tile(ps, bx, 0);
    for (int tid = 0; tid < nt; tid++) ntt_pass_load<E>(lds.data(), tid, in, ps, t);
    int bound = 1;
    for (int st = 0; st < ps.T; st++) {
      const bool weak = ntt_stage_weak(bound);
      // the device replaces the workgroup barrier by wavefront order when ntt_stage_cross is false for
      // both neighbours: run the wavefronts one after the other THROUGH such runs of stages would be the
      // exact emulation; executing stage by stage over all threads is equivalent as long as the butterflies
      // of a local stage stay inside their wavefront's segment, which is asserted here
      for (int tid = 0; tid < E / 2; tid++) {
        const NttBf bf = ntt_bf_index(ps, t, tid, st);
        if (!ntt_stage_cross(ps, st) && ((bf.e0 >> 7) != (tid >> 6) || (bf.e1 >> 7) != (tid >> 6))) fr29_overflows() += 1000;
        ntt_bf_run<E>(lds.data(), bf, ntt_stage_trivial(ps, st), ntt_load_tw(tab + (size_t)bf.pos * NTT_TW), weak);
      }
    }
    for (int tid = 0; tid < nt; tid++) ntt_pass_store<E>(lds.data(), tid, out, tab, ps, t);
  }
}
static bool ntt_host_pass(int loge, const uint32_t* in, uint32_t* out, const uint32_t* tab, const NttPass& ps,
                          std::vector<uint32_t>& lds) {
  switch (loge) {
    case 1: ntt_host_pass_t<1>(in, out, tab, ps, lds); return true;
    case 2: ntt_host_pass_t<2>(in, out, tab, ps, lds); return true;
    case 3: ntt_host_pass_t<3>(in, out, tab, ps, lds); return true;
    case 4: ntt_host_pass_t<4>(in, out, tab, ps, lds); return true;
    case 5: ntt_host_pass_t<5>(in, out, tab, ps, lds); return true;
    case 6: ntt_host_pass_t<6>(in, out, tab, ps, lds); return true;
    case 7: ntt_host_pass_t<7>(in, out, tab, ps, lds); return true;
    case 8: ntt_host_pass_t<8>(in, out, tab, ps, lds); return true;
    case 9: ntt_host_pass_t<9>(in, out, tab, ps, lds); return true;
    case 10: ntt_host_pass_t<10>(in, out, tab, ps, lds); return true;
  }
  return false;
}

// Host twin for the CPU unit tests (hosttest.hip): the SAME pass schedule, tile / twiddle index arithmetic,
// fr29 butterflies and boundary conversions as the kernels, the threads of a block executed one after the
// other.  t0max / tmax shrink the passes so that small transforms exercise the multi-pass paths.  Returns
// the number of 64-bit column / 32-bit limb overflows seen by fr29.hpp's host checks (must be 0).
int ntt_host(int n, const uint32_t* omega_wire, const uint32_t* src, uint32_t* dst, int flags, int t0max, int tmax) {
  const size_t N = (size_t)1 << n;
  if (n == 0) {
    for (int l = 0; l < 8; l++) dst[l] = src[l];
    return 0;
  }
  // the table of ntt_build_table
  std::vector<uint32_t> tab((N + 1) * NTT_TW);
  {
    Fr w;
    for (int i = 0; i < 8; i++) w.v[i] = omega_wire[i];
    w = fp_to_mont<ParamsBlsR>(w);
    const Fr k261 = Fr::from_const(Fr29PR::K261);
    Fr acc = Fr::one();
    for (size_t k = 0; k < N; k++) {
      const Fr e = acc * k261;
      ntt_store_tw(tab.data() + k * NTT_TW, e);
      acc = acc * w;
    }
    Fr ninv = Fr::one(), h = Fr::from_const(ParamsBlsR::INV2);
    for (int i = 0; i < n; i++) ninv = ninv * h;
    ninv = ninv * k261;
    ntt_store_tw(tab.data() + N * NTT_TW, ninv);
  }
  fr29_overflows() = 0;
  const NttSchedule sc = ntt_schedule(n, n, flags, t0max, tmax);
  std::vector<uint32_t> ws(N * 8), dcopy(N * 8), lds;
  for (int k = 0; k < sc.np; k++) {
    const NttPass& ps = sc.ps[k];
    const uint32_t* in = sc.in[k] == 0 ? src : sc.in[k] == 1 ? dcopy.data() : ws.data();
    uint32_t* out = sc.out[k] == 1 ? dcopy.data() : ws.data();
    if (!ntt_host_pass(ps.T + ps.logC, in, out, tab.data(), ps, lds)) return -1;
  }
  for (size_t i = 0; i < N * 8; i++) dst[i] = dcopy[i];
  return fr29_overflows();
}

}  // namespace ncg
