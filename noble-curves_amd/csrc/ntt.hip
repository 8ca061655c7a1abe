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
// Device plan: decimation-in-frequency (natural in -> bit-reversed out) or decimation-in-time
// (bit-reversed in -> natural out) butterflies of the reference's loop (:454-475), grouped into
// passes of up to 8-10 stages that run on an LDS tile (limb-major, conflict-free), one HBM
// round trip per pass.  The remaining bit reversal of the (0,0) and (1,1) forms is folded into
// the last pass's store.  Data stays in canonical form end to end: twiddles are kept in
// Montgomery form, and montmul(a, w R) = a w.  The twiddle table is the reference's natural
// order roots(bits) (N entries); inverse(bits)[k] = roots[(N - k) mod N] is index arithmetic.
#include <vector>

#include "fp.hpp"
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
__global__ void __launch_bounds__(256) k_ntt_pass(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                  const uint32_t* __restrict__ tab, NttPass ps) {
  extern __shared__ uint32_t lds[];
  const int T = ps.T, logC = ps.logC, C = 1 << logC, E = 1 << (T + logC);
  const int L = ps.s_lo - 1;  // index bits below the tile rows
  const size_t N = (size_t)1 << ps.n;
  const uint32_t lochunk = blockIdx.x & ((1u << (L - logC)) - 1u);
  const uint32_t hi = blockIdx.x >> (L - logC);
  const size_t poly = (size_t)blockIdx.y * N * 8;
  const size_t base = ((size_t)hi << (L + T)) | ((size_t)lochunk << logC);

  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const size_t g = base | ((size_t)(e >> logC) << L) | (size_t)(e & (C - 1));
    Fr v = fr_load_g(src + poly + g * 8);
#pragma unroll
    for (int i = 0; i < 8; i++) lds[i * E + e] = v.v[i];
  }
  __syncthreads();

  const uint32_t tmask = (uint32_t)((N << ps.tshift) - 1);
  for (int st = 0; st < T; st++) {
    const int q = ps.dit ? st : T - 1 - st;  // tile-row bit split by this stage
    const int s = ps.s_lo + q;               // FFTCore stage: m = 2^s, stride = N >> s  (fft.ts:456-460)
    for (int b = threadIdx.x; b < E / 2; b += blockDim.x) {
      const uint32_t c = b & (C - 1), bm = b >> logC;
      const uint32_t low = bm & ((1u << q) - 1u);
      const uint32_t mid0 = ((bm >> q) << (q + 1)) | low;
      const int e0 = (mid0 << logC) | c, e1 = e0 + (1 << (q + logC));
      // j = i0 mod m/2; rootPos = j * (N >> s)  (fft.ts:463-467)
      const uint32_t j = (low << L) | (lochunk << logC) | c;
      uint32_t pos = (j << (ps.n - s)) << ps.tshift;
      if (ps.inverse) pos = (0u - pos) & tmask;  // inverse(bits)[k] = roots[(N - k) mod N]  (fft.ts:296-304)
      const Fr w = fr_load_g(tab + (size_t)pos * 8);
      Fr a, bb;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        a.v[i] = lds[i * E + e0];
        bb.v[i] = lds[i * E + e1];
      }
      Fr o0, o1;
      if (ps.dit) {  // fft.ts:470-473
        Fr t = bb * w;
        o0 = a + t;
        o1 = a - t;
      } else {  // fft.ts:476-478
        o0 = a + bb;
        o1 = (a - bb) * w;
      }
#pragma unroll
      for (int i = 0; i < 8; i++) {
        lds[i * E + e0] = o0.v[i];
        lds[i * E + e1] = o1.v[i];
      }
    }
    __syncthreads();
  }

  Fr ninv;
  if (ps.scale) ninv = fr_load_g(tab + (((size_t)N << ps.tshift)) * 8);  // slot after the table: (1/N) R
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    size_t g = base | ((size_t)(e >> logC) << L) | (size_t)(e & (C - 1));
    Fr v;
#pragma unroll
    for (int i = 0; i < 8; i++) v.v[i] = lds[i * E + e];
    if (ps.scale) v = v * ninv;  // fft.ts:568-570
    if (ps.brp_store) g = ps.n ? (size_t)(__brev((uint32_t)g) >> (32 - ps.n)) : 0;
    fr_store_g(dst + poly + g * 8, v);
  }
}

// ---- twiddle table: tab[k] = omega^k (Montgomery form), k < N; tab[N] = (1/N) (Montgomery form)
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
    fr_store_g(tab + k * 8, r);
    return;
  }
  const Fr a = fr_load_g(small + (k & (((size_t)1 << lo) - 1)) * 8);
  const Fr b = fr_load_g(small + (((size_t)1 << lo) + (k >> lo)) * 8);
  fr_store_g(tab + k * 8, a * b);
}

size_t ntt_table_bytes(int n) { return (((size_t)1 << n) + 1) * 32; }

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

// Stage groups, lowest first: the lowest covers up to 10 stages on contiguous 2^T tiles, the rest
// are split evenly into groups of at most 8 stages on 2^T x 4 tiles.
int ntt_plan(int n, int (&s_lo)[8], int (&T)[8]) {
  if (n == 0) return 0;
  int np = 0;
  const int t0 = n < 10 ? n : 10;
  s_lo[np] = 1;
  T[np++] = t0;
  int rem = n - t0, s = t0 + 1;
  const int groups = (rem + 7) / 8;
  for (int g = 0; g < groups; g++) {
    int t = (rem + (groups - g) - 1) / (groups - g);
    s_lo[np] = s;
    T[np++] = t;
    s += t;
    rem -= t;
  }
  return np;
}

// flags: bit 0 inverse, bit 1 brpInput, bit 2 brpOutput.  ws: batch * N * 32 bytes (only read when
// the bit reversal is folded into a multi-pass transform); src may equal dst.
hipError_t ntt_run(int n, size_t batch, const uint32_t* src, uint32_t* dst, uint32_t* ws, const uint32_t* tab,
                   int tab_log, int flags, hipStream_t st) {
  const bool inverse = flags & 1, brp_in = flags & 2, brp_out = flags & 4;
  const size_t N = (size_t)1 << n;
  int s_lo[8], T[8];
  const int np = ntt_plan(n, s_lo, T);
  if (np == 0) {  // N = 1: identity (and 1/N = 1)
    if (src != dst) return hipMemcpyAsync(dst, src, batch * 32, hipMemcpyDeviceToDevice, st);
    return hipSuccess;
  }
  const bool dit = brp_in;
  const bool fold_brp = brp_in == brp_out;  // (0,0): DIF then BR;  (1,1): DIT then BR
  for (int k = 0; k < np; k++) {
    const int g = dit ? k : np - 1 - k;  // DIT ascends the stages, DIF descends
    NttPass ps;
    ps.n = n;
    ps.s_lo = s_lo[g];
    ps.T = T[g];
    ps.logC = s_lo[g] == 1 ? 0 : 2;
    ps.dit = dit;
    ps.inverse = inverse;
    ps.brp_store = fold_brp && k == np - 1;
    ps.scale = inverse && k == np - 1;
    ps.tshift = tab_log - n;
    const uint32_t* in;
    uint32_t* out;
    if (fold_brp && np > 1) {
      in = k == 0 ? src : ws;
      out = k == np - 1 ? dst : ws;
    } else {
      in = k == 0 ? src : dst;
      out = dst;
    }
    const int E = 1 << (ps.T + ps.logC);
    const unsigned tiles = (unsigned)(N >> (ps.T + ps.logC));
    int threads = E / 2 < 256 ? (E / 2 < 64 ? 64 : E / 2) : 256;
    hipLaunchKernelGGL(k_ntt_pass, dim3(tiles, (unsigned)batch), dim3(threads), (size_t)E * 32, st, in, out, tab, ps);
  }
  return hipGetLastError();
}

// host-side reference of the pass structure for the CPU unit tests (hosttest.hip): the same
// butterflies in the same grouping, executed serially.
void ntt_host(int n, const uint32_t* omega_wire, const uint32_t* src, uint32_t* dst, int flags) {
  const bool inverse = flags & 1, brp_in = flags & 2, brp_out = flags & 4;
  const size_t N = (size_t)1 << n;
  std::vector<Fr> tab(N + 1), v(N);
  Fr w;
  for (int i = 0; i < 8; i++) w.v[i] = omega_wire[i];
  w = fp_to_mont<ParamsBlsR>(w);
  tab[0] = Fr::one();
  for (size_t k = 1; k < N; k++) tab[k] = tab[k - 1] * w;
  Fr ninv = Fr::one(), h = Fr::from_const(ParamsBlsR::INV2);
  for (int i = 0; i < n; i++) ninv = ninv * h;
  for (size_t i = 0; i < N; i++)
    for (int l = 0; l < 8; l++) v[i].v[l] = src[i * 8 + l];
  const bool dit = brp_in;
  for (int st = 0; st < n; st++) {
    const int s = dit ? st + 1 : n - st;
    const size_t m2 = (size_t)1 << (s - 1);
    for (size_t i0 = 0; i0 < N; i0++) {
      if (i0 & m2) continue;
      const size_t i1 = i0 | m2, j = i0 & (m2 - 1);
      size_t pos = j << (n - s);
      if (inverse) pos = (N - pos) & (N - 1);
      const Fr a = v[i0], b = v[i1];
      if (dit) {
        Fr t = b * tab[pos];
        v[i0] = a + t;
        v[i1] = a - t;
      } else {
        v[i0] = a + b;
        v[i1] = (a - b) * tab[pos];
      }
    }
  }
  for (size_t i = 0; i < N; i++) {
    Fr x = inverse ? v[i] * ninv : v[i];
    size_t g = i;
    if (brp_in == brp_out && n) {
      size_t r = 0;
      for (int b = 0; b < n; b++) r |= ((i >> b) & 1) << (n - 1 - b);
      g = r;
    }
    for (int l = 0; l < 8; l++) dst[g * 8 + l] = x.v[l];
  }
}

}  // namespace ncg
