// Go / no-go micro-benchmark (VERDICT r04 "next" #5b): a 13-limb radix-2^30 Montgomery product for Fp381 (169 + 169
// v_mad_u64_u32) against the shipped 14-limb radix-2^29 form (196 + 196).
//
// Both are the same row-wise interleaved algorithm on 64-bit columns (fp29.hpp's shape): row i adds a_i * b and q_i * p into
// columns i .. i + L - 1, then hands the carry of column i to column i + 1.  What differs is HEADROOM: a column receives up to
// 2 L products.  With 29-bit limbs that is 28 x 2^58 = 2^62.8 - no column ever overflows, and one operand may even carry a lazy
// addition (limbs up to 2^30).  With 30-bit limbs it is 26 x 2^60 = 2^64.7: the middle columns overflow unless they are
// carry-normalised halfway (here: after row 6, every live column hands its upper bits on - 64-bit shift, mask, 64-bit add per
// column), and BOTH operands must hold fully normalised limbs, so every lazy add / sub of the curve formulas in front of a
// product needs its own carry pass (13 x 3 plain operations; not timed here - it only adds to the 30-bit side).
// Each product is checked against a host big-integer Montgomery product; then both loops are timed at the same occupancy.
// Measurement tooling for DESIGN.md section 3; not on the product path.
//   hipcc --offload-arch=gfx950 -O3 tools/mont_radix_ubench.hip -o tools/_build/mont_radix_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITERS = 256;

template <int L, int B> struct Num { uint32_t v[L]; };

// result = a * b * 2^(-L B) mod p, limbs below 2^B, value below 2 p (no final subtraction: as in the kernels)
template <int L, int B, bool MID>
__device__ __forceinline__ Num<L, B> mont(const Num<L, B>& a, const Num<L, B>& b, const uint32_t* __restrict__ p, uint32_t n0inv) {
  constexpr uint32_t M = (1u << B) - 1u;
  uint64_t t[2 * L + 1];
#pragma unroll
  for (int k = 0; k < 2 * L + 1; k++) t[k] = 0;
#pragma unroll
  for (int i = 0; i < L; i++) {
#pragma unroll
    for (int j = 0; j < L; j++) t[i + j] += (uint64_t)a.v[i] * b.v[j];
    const uint32_t q = ((uint32_t)t[i] * n0inv) & M;
#pragma unroll
    for (int j = 0; j < L; j++) t[i + j] += (uint64_t)q * p[j];
    t[i + 1] += t[i] >> B;
    if constexpr (MID) {
      if (i == L / 2) {   // the live columns hand their upper bits on before the second half of the rows arrives
#pragma unroll
        for (int k = i + 1; k < i + L; k++) {
          t[k + 1] += t[k] >> B;
          t[k] &= M;
        }
      }
    }
  }
  Num<L, B> r;
#pragma unroll
  for (int k = 0; k < L; k++) {
    r.v[k] = (uint32_t)t[L + k] & M;
    t[L + k + 1] += t[L + k] >> B;
  }
  return r;
}

template <int L, int B, bool MID>
__global__ void __launch_bounds__(256) k_mont(const uint32_t* __restrict__ xs, const uint32_t* __restrict__ pl, uint32_t n0inv,
                                              uint32_t* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  Num<L, B> a, b;
  uint32_t p[L];
#pragma unroll
  for (int i = 0; i < L; i++) {
    a.v[i] = xs[(size_t)t * 2 * L + i];
    b.v[i] = xs[(size_t)t * 2 * L + L + i];
    p[i] = pl[i];
  }
  for (int it = 0; it < iters; it++) {
    const Num<L, B> c = mont<L, B, MID>(a, b, p, n0inv);
    b = a;
    a = c;
  }
#pragma unroll
  for (int i = 0; i < L; i++) out[(size_t)t * L + i] = a.v[i];
}

// ---------------------------------------------------------------- host big integers (base 2^32, 16 words)
typedef unsigned __int128 u128;
struct Big { uint32_t w[32]; };
static Big big_zero() { Big z; memset(&z, 0, sizeof z); return z; }
static int big_cmp(const Big& a, const Big& b) { for (int i = 31; i >= 0; i--) if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1; return 0; }
static Big big_sub(const Big& a, const Big& b) { Big r; int64_t bw = 0; for (int i = 0; i < 32; i++) { int64_t t = (int64_t)a.w[i] - b.w[i] - bw; bw = t < 0; r.w[i] = (uint32_t)t; } return r; }
static Big big_mul(const Big& a, const Big& b) { Big r = big_zero(); for (int i = 0; i < 16; i++) { uint64_t cy = 0; for (int j = 0; j + i < 32 && j < 16; j++) { u128 t = (u128)a.w[i] * b.w[j] + r.w[i + j] + cy; r.w[i + j] = (uint32_t)t; cy = (uint64_t)(t >> 32); } if (i + 16 < 32) r.w[i + 16] += (uint32_t)cy; } return r; }
static Big big_mod(Big a, const Big& p) {   // shift-subtract, a < 2^832
  for (int sh = 832 - 381; sh >= 0; sh--) {
    Big ps = big_zero();
    for (int i = 0; i < 32; i++) {
      const int src = i - sh / 32;
      uint64_t v = 0;
      if (src >= 0 && src < 32) v = (uint64_t)p.w[src] << (sh % 32);
      if (src - 1 >= 0 && src - 1 < 32 && sh % 32) v |= (uint64_t)p.w[src - 1] >> (32 - sh % 32);
      ps.w[i] = (uint32_t)v;
    }
    if (big_cmp(a, ps) >= 0) a = big_sub(a, ps);
  }
  return a;
}
template <int L, int B> static Big from_limbs(const uint32_t* v) {
  Big r = big_zero();
  for (int i = 0; i < L; i++) {
    const int bit = B * i;
    const uint64_t t = (uint64_t)v[i] << (bit % 32);
    r.w[bit / 32] |= (uint32_t)t;   // limbs below 2^B never overlap
    r.w[bit / 32 + 1] |= (uint32_t)(t >> 32);
  }
  return r;
}
template <int L, int B> static void to_limbs(const Big& a, uint32_t* v) {
  for (int i = 0; i < L; i++) {
    const int bit = B * i;
    const uint64_t t = ((uint64_t)a.w[bit / 32 + 1] << 32) | a.w[bit / 32];
    v[i] = (uint32_t)(t >> (bit % 32)) & ((1u << B) - 1u);
  }
}

template <int L, int B, bool MID>
static int run(const char* name, const Big& P, int waves_per_simd, int cus) {
  const int blocks = cus * waves_per_simd, threads = blocks * 256;
  uint32_t pl[L];
  to_limbs<L, B>(P, pl);
  uint32_t n0 = 1;   // -p^-1 mod 2^B by Newton
  for (int i = 0; i < 6; i++) n0 *= 2u - pl[0] * n0;
  const uint32_t n0inv = (0u - n0) & ((1u << B) - 1u);
  std::vector<uint32_t> xs((size_t)threads * 2 * L);
  uint64_t s = 0x9e3779b97f4a7c15ull + B;
  for (int t = 0; t < threads; t++)
    for (int h = 0; h < 2; h++) {
      Big x = big_zero();
      for (int i = 0; i < 12; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x.w[i] = (uint32_t)s; }
      x.w[11] &= 0x0fffffffu;   // below 2^380 < p
      to_limbs<L, B>(x, &xs[((size_t)t * 2 + h) * L]);
    }
  uint32_t *d_x, *d_p, *d_o;
  CK(hipMalloc(&d_x, xs.size() * 4));
  CK(hipMalloc(&d_p, L * 4));
  CK(hipMalloc(&d_o, (size_t)threads * L * 4));
  CK(hipMemcpy(d_x, xs.data(), xs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_p, pl, L * 4, hipMemcpyHostToDevice));
  // check: one product of the first 256 lanes: got * 2^(L B) == a * b (mod p)
  hipLaunchKernelGGL((k_mont<L, B, MID>), dim3(blocks), dim3(256), 0, 0, d_x, d_p, n0inv, d_o, 1);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> got((size_t)threads * L);
  CK(hipMemcpy(got.data(), d_o, got.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  Big R = big_zero();
  R.w[(L * B) / 32] = 1u << ((L * B) % 32);
  R = big_mod(R, P);
  for (int t = 0; t < 256; t++) {
    const Big a = from_limbs<L, B>(&xs[(size_t)t * 2 * L]), b = from_limbs<L, B>(&xs[(size_t)t * 2 * L + L]);
    const Big want = big_mod(big_mul(a, b), P);
    const Big have = big_mod(big_mul(big_mod(from_limbs<L, B>(&got[(size_t)t * L]), P), R), P);
    if (big_cmp(want, have) != 0) bad++;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mont<L, B, MID>), dim3(blocks), dim3(256), 0, 0, d_x, d_p, n0inv, d_o, ITERS);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  printf("{\"form\": \"%s\", \"limbs\": %d, \"radix_bits\": %d, \"mads_per_product\": %d, \"waves_per_simd\": %d, \"mismatches\": %d, "
         "\"kernel_ms\": %.4f, \"cycles_per_wave_product_at_2.4GHz\": %.0f, \"products_per_s\": %.3e}\n",
         name, L, B, 2 * L * L, waves_per_simd, bad, best, best * 1e-3 * 2.4e9 / ((double)ITERS * waves_per_simd),
         (double)threads * ITERS / (best * 1e-3));
  (void)hipFree(d_x); (void)hipFree(d_p); (void)hipFree(d_o);
  return bad;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  static const char* P_HEX = "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab";
  Big P = big_zero();
  const int n = (int)strlen(P_HEX);
  for (int i = 0; i < n / 8; i++) {
    unsigned v;
    sscanf(P_HEX + n - 8 - 8 * i, "%8x", &v);
    P.w[i] = v;
  }
  int bad = 0;
  for (int W : {1, 2, 4}) {
    if (argc > 1 && atoi(argv[1]) != W) continue;
    bad += run<14, 29, false>("radix 2^29, 14 limbs (shipped shape)", P, W, prop.multiProcessorCount);
    bad += run<13, 30, true>("radix 2^30, 13 limbs, columns normalised halfway", P, W, prop.multiProcessorCount);
  }
  return bad ? 2 : 0;
}
