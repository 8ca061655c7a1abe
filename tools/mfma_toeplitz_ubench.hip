// Exploratory micro-benchmark (VERDICT r04 "next" #6): can the idle matrix pipe take the CONSTANT half of a Montgomery product?
//
// The reduction half of fp29.hpp's product multiplies by constants (t * N', q * p): for a batch of 64 lanes that is a batch of
// digit vectors times a fixed Toeplitz matrix - a dense int8 contraction, and MFMA issues beside the VALU.  This tool times ONE
// multiplication of 64 Fp381-sized values (14 limbs of 29 bits, one value per lane - the layout every kernel of libncg keeps)
// by a 406-bit constant, both ways, INCLUDING what the matrix route needs around the MFMAs:
//   valu:  196 v_mad_u64_u32 into 64-bit columns + carry extraction (the shape of the product / reduction halves in fp29.hpp)
//   mfma:  limbs -> 13 packed dwords (radix 2^29 -> 2^32), signed-byte recoding (+0x80 per byte with carries, xor 0x80),
//          transpose through LDS into the A-operand layout (lane = row + 16 * k-block, 16 bytes each), 4 x NT
//          v_mfma_i32_16x16x64_i8 against the constant's Toeplitz tiles (B operands resident in registers), transpose of the
//          i32 column sums back through LDS (column-major), 64-bit recombination of the byte columns, carry propagation,
//          radix 2^32 -> 2^29.
// Both results are compared limb for limb on the host side of this file (and with each other), then each loop is timed at the same
// occupancy.  NT = 7 is the full 102-column product; NT = 4 (64 columns) is the lower bound for a reduction that only needs
// the high half plus a few guard columns (its output is not checked, it only bounds the cost).
// Measurement tooling for DESIGN.md section 3; not on the product path.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_toeplitz_ubench.hip -o tools/_build/mfma_toeplitz_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int NL = 14, LB = 29;
constexpr uint32_t LM = (1u << LB) - 1u;
constexpr int NDW = 13;             // 13 dwords = 52 bytes >= 406 bits (+ 1 recoding carry)
constexpr int NTMAX = 7;            // 7 column tiles of 16 = 112 >= 52 + 51 - 1 columns
constexpr int ITERS = 64;

struct Limbs { uint32_t v[NL]; };
struct Prod { uint32_t v[2 * NL]; };

// ---------------------------------------------------------------- VALU: schoolbook in 64-bit columns
__device__ __forceinline__ Prod mul_valu(const Limbs& a, const uint32_t* __restrict__ c) {
  uint64_t col[2 * NL - 1];
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) col[k] = 0;
#pragma unroll
  for (int i = 0; i < NL; i++)
#pragma unroll
    for (int j = 0; j < NL; j++) col[i + j] += (uint64_t)a.v[i] * c[j];
  Prod p;
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
    const uint64_t t = col[k] + carry;
    p.v[k] = (uint32_t)t & LM;
    carry = t >> LB;
  }
  p.v[2 * NL - 1] = (uint32_t)carry;
  return p;
}

// ---------------------------------------------------------------- MFMA route
// 14 limbs of 29 bits -> 13 little-endian dwords
__device__ __forceinline__ void limbs_to_dwords(const Limbs& a, uint32_t* w) {
#pragma unroll
  for (int d = 0; d < NDW; d++) {
    const int bit = 32 * d, i = bit / LB, s = bit % LB;
    uint64_t t = (uint64_t)(i < NL ? a.v[i] : 0u) >> s;
    if (i + 1 < NL) t |= (uint64_t)a.v[i + 1] << (LB - s);
    if (i + 2 < NL && 2 * LB - s < 32) t |= (uint64_t)a.v[i + 2] << (2 * LB - s);
    w[d] = (uint32_t)t;
  }
}
// x = sum_k s_k 2^(8k) with s_k in [-128, 127]: add 0x80 to bytes 0..50 (with carries), flip their top bits; byte 51 = carry (0 / 1)
__device__ __forceinline__ void recode_signed_bytes(uint32_t* w) {
  uint32_t cy = 0;
#pragma unroll
  for (int d = 0; d < NDW; d++) {
    const uint32_t k = d < NDW - 1 ? 0x80808080u : 0x00808080u;
    const uint64_t t = (uint64_t)w[d] + k + cy;
    w[d] = (uint32_t)t ^ k;
    cy = (uint32_t)(t >> 32);
  }
}

template <int NT, bool CHECKED>
__device__ __forceinline__ Prod mul_mfma(const Limbs& a, const v4i* __restrict__ bfrag, uint32_t* lds_a, int32_t* lds_c) {
  const int lane = threadIdx.x & 63;
  uint32_t w[16];
  limbs_to_dwords(a, w);
  recode_signed_bytes(w);
  w[13] = w[14] = w[15] = 0;
  // element-major: 16 dwords per element
  uint4* mine = reinterpret_cast<uint4*>(lds_a + lane * 16);
#pragma unroll
  for (int j = 0; j < 4; j++) mine[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are ordered before its reads
  __builtin_amdgcn_wave_barrier();
  const int r = lane & 15, kb = lane >> 4;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const uint4 av = *reinterpret_cast<const uint4*>(lds_a + (16 * g + r) * 16 + 4 * kb);
    const v4i A = {(int)av.x, (int)av.y, (int)av.z, (int)av.w};
#pragma unroll
    for (int ct = 0; ct < NT; ct++) {
      const v4i z = {0, 0, 0, 0};
      const v4i D = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bfrag[ct], z, 0, 0, 0);
      // D: column 16 ct + r of elements 16 g + 4 kb + (0..3)  ->  column-major [col][elem]
      *reinterpret_cast<v4i*>(lds_c + (16 * ct + r) * 64 + 16 * g + 4 * kb) = D;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  // this lane's element: byte columns -> 64-bit words of 32 bits each (signed), carries, then 29-bit limbs
  constexpr int NCOL = 16 * NT;
  constexpr int NW = NCOL / 4;
  int64_t word[NW];
#pragma unroll
  for (int d = 0; d < NW; d++) {
    int64_t t = 0;
#pragma unroll
    for (int b = 3; b >= 0; b--) t = (t << 8) + (int64_t)lds_c[(4 * d + b) * 64 + lane];
    word[d] = t;
  }
  uint32_t u[NW + 1];
  int64_t cy = 0;
#pragma unroll
  for (int d = 0; d < NW; d++) {
    const int64_t t = word[d] + cy;
    u[d] = (uint32_t)t;
    cy = t >> 32;
  }
  u[NW] = (uint32_t)cy;
  Prod p;
#pragma unroll
  for (int k = 0; k < 2 * NL; k++) {
    const int bit = LB * k, d = bit / 32, s = bit % 32;
    uint64_t t = 0;
    if (d <= NW) t = (uint64_t)u[d] >> s;
    if (d + 1 <= NW) t |= (uint64_t)u[d + 1] << (32 - s);
    p.v[k] = (uint32_t)t & LM;
  }
  (void)CHECKED;
  return p;
}

// MODE 0: VALU, 1: MFMA (full, NT = 7), 2: MFMA (NT = 4: lower bound for a high-half-only reduction)
template <int MODE>
__global__ void __launch_bounds__(256) k_bench(const uint32_t* __restrict__ xs, const uint32_t* __restrict__ cl, const v4i* __restrict__ bfr,
                                               uint32_t* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_a[4][64 * 16];
  __shared__ __attribute__((aligned(16))) int32_t lds_c[4][16 * NTMAX * 64];
  const int t = blockIdx.x * blockDim.x + threadIdx.x, wv = threadIdx.x >> 6;
  Limbs x;
#pragma unroll
  for (int i = 0; i < NL; i++) x.v[i] = xs[(size_t)t * NL + i];
  uint32_t c[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) c[i] = cl[i];
  v4i bf[NTMAX];
#pragma unroll
  for (int ct = 0; ct < NTMAX; ct++) bf[ct] = bfr[ct * 64 + (threadIdx.x & 63)];
  Prod p;
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) p = mul_valu(x, c);
    else if constexpr (MODE == 1) p = mul_mfma<7, true>(x, bf, lds_a[wv], lds_c[wv]);
    else p = mul_mfma<4, false>(x, bf, lds_a[wv], lds_c[wv]);
    if (it + 1 < iters) {   // the next input depends on this product (nothing hoists, nothing overlaps across iterations for free)
#pragma unroll
      for (int i = 0; i < NL; i++) x.v[i] = (p.v[i] ^ p.v[NL + i]) & LM;
      x.v[NL - 1] &= (1u << (406 - LB * (NL - 1))) - 1u;   // keep the value below 2^406
    }
  }
#pragma unroll
  for (int k = 0; k < 2 * NL; k++) out[(size_t)t * 2 * NL + k] = p.v[k];
}

// ---------------------------------------------------------------- host: constant tiles, reference product, timing
static void host_mul(const uint32_t* a, const uint32_t* c, uint32_t* p) {
  unsigned __int128 col[2 * NL - 1];
  for (auto& v : col) v = 0;
  for (int i = 0; i < NL; i++)
    for (int j = 0; j < NL; j++) col[i + j] += (unsigned __int128)a[i] * c[j];
  unsigned __int128 cy = 0;
  for (int k = 0; k < 2 * NL - 1; k++) {
    unsigned __int128 t = col[k] + cy;
    p[k] = (uint32_t)t & LM;
    cy = t >> LB;
  }
  p[2 * NL - 1] = (uint32_t)cy;
}
static void host_next(const uint32_t* p, uint32_t* x) {
  for (int i = 0; i < NL; i++) x[i] = (p[i] ^ p[NL + i]) & LM;
  x[NL - 1] &= (1u << (406 - LB * (NL - 1))) - 1u;
}

int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * waves_per_simd;   // 4 waves per block = one per SIMD; `waves_per_simd` blocks per CU
  const int threads = blocks * 256;
  // the constant: the bls12-381 base-field prime (the q * p half), 29-bit limbs
  static const char* P_HEX = "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab";
  uint8_t pb[64] = {0};   // little-endian bytes
  {
    const int n = (int)strlen(P_HEX);
    for (int i = 0; i < n / 2; i++) {
      unsigned v;
      sscanf(P_HEX + n - 2 - 2 * i, "%2x", &v);
      pb[i] = (uint8_t)v;
    }
  }
  uint32_t cl[NL];
  for (int i = 0; i < NL; i++) {
    uint64_t t = 0;
    for (int b = 0; b < 8; b++) {
      const int byte = (LB * i) / 8 + b;
      if (byte < 64) t |= (uint64_t)pb[byte] << (8 * b);
    }
    cl[i] = (uint32_t)(t >> ((LB * i) % 8)) & LM;
  }
  // signed-byte digits of the constant (same recoding as on the device: +0x80 per byte, carries, xor)
  int8_t cs[64] = {0};
  {
    unsigned cy = 0;
    for (int k = 0; k < 52; k++) {
      const unsigned kk = k < 51 ? 0x80u : 0u;
      const unsigned t = pb[k] + kk + cy;
      cs[k] = (int8_t)((t & 0xffu) ^ kk);
      cy = t >> 8;
    }
  }
  // B tile ct, lane l: column n = l & 15, k-block l >> 4: bytes B[k][n] = cs[16 ct + n - k] for k = 16 (l >> 4) + 0..15
  std::vector<v4i> bfr(NTMAX * 64);
  for (int ct = 0; ct < NTMAX; ct++)
    for (int l = 0; l < 64; l++) {
      int8_t by[16];
      for (int i = 0; i < 16; i++) {
        const int k = 16 * (l >> 4) + i, j = 16 * ct + (l & 15) - k;
        by[i] = (j >= 0 && j < 52 && k < 52) ? cs[j] : 0;
      }
      memcpy(&bfr[ct * 64 + l], by, 16);
    }
  std::vector<uint32_t> xs((size_t)threads * NL);
  uint64_t s = 0x9e3779b97f4a7c15ull;
  for (auto& v : xs) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    v = (uint32_t)s & LM;
  }
  for (int t = 0; t < threads; t++) xs[(size_t)t * NL + NL - 1] &= (1u << (406 - LB * (NL - 1))) - 1u;
  uint32_t *d_x, *d_c, *d_out;
  v4i* d_b;
  CK(hipMalloc(&d_x, xs.size() * 4));
  CK(hipMalloc(&d_c, NL * 4));
  CK(hipMalloc(&d_b, bfr.size() * sizeof(v4i)));
  CK(hipMalloc(&d_out, (size_t)threads * 2 * NL * 4));
  CK(hipMemcpy(d_x, xs.data(), xs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_c, cl, NL * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, bfr.data(), bfr.size() * sizeof(v4i), hipMemcpyHostToDevice));
  // correctness: 3 chained products of the first 512 lanes against the host, both routes
  std::vector<uint32_t> got((size_t)threads * 2 * NL);
  int bad[2] = {0, 0};
  for (int mode = 0; mode < 2; mode++) {
    if (mode == 0) hipLaunchKernelGGL(k_bench<0>, dim3(blocks), dim3(256), 0, 0, d_x, d_c, d_b, d_out, 3);
    else hipLaunchKernelGGL(k_bench<1>, dim3(blocks), dim3(256), 0, 0, d_x, d_c, d_b, d_out, 3);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    for (int t = 0; t < 512; t++) {
      uint32_t x[NL], p[2 * NL];
      memcpy(x, &xs[(size_t)t * NL], NL * 4);
      for (int it = 0; it < 3; it++) {
        host_mul(x, cl, p);
        if (it < 2) host_next(p, x);
      }
      if (memcmp(p, &got[(size_t)t * 2 * NL], 2 * NL * 4) != 0) bad[mode]++;
    }
  }
  printf("{\"check\": {\"valu_mismatches\": %d, \"mfma_mismatches\": %d, \"lanes_checked\": 512}}\n", bad[0], bad[1]);
  // timing
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[3] = {"valu_196_mad", "mfma_i8_full_102_columns", "mfma_i8_64_columns_lower_bound"};
  double ms_mode[3];
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k_bench<0>, dim3(blocks), dim3(256), 0, 0, d_x, d_c, d_b, d_out, ITERS);
      else if (mode == 1) hipLaunchKernelGGL(k_bench<1>, dim3(blocks), dim3(256), 0, 0, d_x, d_c, d_b, d_out, ITERS);
      else hipLaunchKernelGGL(k_bench<2>, dim3(blocks), dim3(256), 0, 0, d_x, d_c, d_b, d_out, ITERS);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    ms_mode[mode] = best;
    // cycles one SIMD spends per wave-product: time * clock / (iters * waves per SIMD)
    const double cyc = best * 1e-3 * 2.4e9 / ((double)ITERS * waves_per_simd);
    printf("{\"route\": \"%s\", \"waves_per_simd\": %d, \"kernel_ms\": %.4f, \"cycles_per_wave_product_at_2.4GHz\": %.0f, \"products_per_s\": %.3e}\n",
           names[mode], waves_per_simd, best, cyc, (double)threads * ITERS / (best * 1e-3));
  }
  printf("{\"summary\": \"constant multiply of 64 x 406-bit values: mfma/valu time ratio %.2f (full), %.2f (64-column lower bound)\"}\n",
         ms_mode[1] / ms_mode[0], ms_mode[2] / ms_mode[0]);
  return (bad[0] || bad[1]) ? 2 : 0;
}
