// Issue-cost measurement of the VALU instructions the field arithmetic is made of (gfx950).
// Every kernel runs a loop of UNROLL identical inline-asm instructions on independent registers
// (or one dependent chain for the latency rows), W waves per SIMD, and brackets the loop with
// s_memtime; the host reports shader cycles per wave-instruction per SIMD = elapsed / (N * W).
// Measurement tooling for DESIGN.md section 3; not on the product path.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/_build/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ITERS = 512;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// KIND selects the instruction; 8 independent register sets r[0..7] (64-bit where needed)
template <int KIND>
__global__ void __launch_bounds__(256) k_rate(uint64_t* __restrict__ tstamps, uint32_t* __restrict__ sink, uint32_t seed) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t r[8];
  uint32_t a = seed * 2654435761u + t, b = (seed ^ 0x9e3779b9u) + 3u * t, c = t | 1u;
  uint32_t x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r[i] = ((uint64_t)(a + i) << 32) | (b + 7 * i);
    x[i] = a * (i + 1) + b;
  }
  uint64_t t0 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
    if constexpr (KIND == 0) {  // v_mad_u64_u32, 8 independent accumulators
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 1) {  // v_mad_u64_u32, one dependent chain
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[0]) : "v"(a), "v"(b) : "vcc");
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 2) {  // v_add_u32
#define X(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 3) {  // carry chain: add_co then 7 addc (bignum add shape)
      asm volatile(
          "v_add_co_u32 %0, vcc, %8, %0\n\tv_addc_co_u32 %1, vcc, %8, %1, vcc\n\tv_addc_co_u32 %2, vcc, %8, %2, vcc\n\t"
          "v_addc_co_u32 %3, vcc, %8, %3, vcc\n\tv_addc_co_u32 %4, vcc, %8, %4, vcc\n\tv_addc_co_u32 %5, vcc, %8, %5, vcc\n\t"
          "v_addc_co_u32 %6, vcc, %8, %6, vcc\n\tv_addc_co_u32 %7, vcc, %8, %7, vcc\n\t"
          "v_add_co_u32 %0, vcc, %9, %0\n\tv_addc_co_u32 %1, vcc, %9, %1, vcc\n\tv_addc_co_u32 %2, vcc, %9, %2, vcc\n\t"
          "v_addc_co_u32 %3, vcc, %9, %3, vcc\n\tv_addc_co_u32 %4, vcc, %9, %4, vcc\n\tv_addc_co_u32 %5, vcc, %9, %5, vcc\n\t"
          "v_addc_co_u32 %6, vcc, %9, %6, vcc\n\tv_addc_co_u32 %7, vcc, %9, %7, vcc"
          : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
          : "v"(a), "v"(b)
          : "vcc");
    } else if constexpr (KIND == 4) {  // mac96 pattern: mad + addc of its carry (16 instructions)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(r[i]), "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
      REP8(X)
#undef X
    } else if constexpr (KIND == 5) {  // v_mul_lo_u32
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 6) {  // v_mul_hi_u32
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 7) {  // v_lshrrev_b64
#define X(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(r[i]));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 8) {  // v_lshl_add_u64
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 9) {  // v_and_b32
#define X(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 10) {  // v_cndmask_b32 (vcc fixed)
      asm volatile("v_cmp_gt_u32 vcc, %0, %1" ::"v"(a), "v"(b) : "vcc");
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 11) {  // v_add3_u32
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 12) {  // v_mad_u32_u24
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 13) {  // v_mad_i64_i32
#define X(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 14) {  // v_alignbit_b32
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(x[i]) : "v"(a));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 15) {  // v_mov_b32 dpp quad_perm (pair swap)
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 16) {  // v_sub_u32 + v_add_u32 pair (lazy subtraction limb)
#define X(i) asm volatile("v_sub_u32 %0, %0, %1\n\tv_add_u32 %0, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
      REP8(X)
#undef X
    } else if constexpr (KIND == 17) {  // column step of the carry-free radix: 4 mads into one accumulator, shift, mask
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %3, %0\n\tv_and_b32 %1, 0x1fffffff, %1\n\tv_lshrrev_b64 %0, 29, %0" : "+v"(r[i]), "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
      REP8(X)
#undef X
    } else if constexpr (KIND == 18) {  // v_mul_u32_u24
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 20) {  // v_fma_f64 (the DFMA big-integer product of Emmart et al. would be made of these)
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 21) {  // v_add_f64
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == 19) {  // v_subb_co_u32 chain + cndmask (conditional subtract shape)
      asm volatile(
          "v_sub_co_u32 %0, vcc, %0, %8\n\tv_subb_co_u32 %1, vcc, %1, %8, vcc\n\tv_subb_co_u32 %2, vcc, %2, %8, vcc\n\t"
          "v_subb_co_u32 %3, vcc, %3, %8, vcc\n\tv_subb_co_u32 %4, vcc, %4, %8, vcc\n\tv_subb_co_u32 %5, vcc, %5, %8, vcc\n\t"
          "v_subb_co_u32 %6, vcc, %6, %8, vcc\n\tv_subb_co_u32 %7, vcc, %7, %8, vcc\n\t"
          "v_cndmask_b32 %0, %0, %9, vcc\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cndmask_b32 %2, %2, %9, vcc\n\t"
          "v_cndmask_b32 %3, %3, %9, vcc\n\tv_cndmask_b32 %4, %4, %9, vcc\n\tv_cndmask_b32 %5, %5, %9, vcc\n\t"
          "v_cndmask_b32 %6, %6, %9, vcc\n\tv_cndmask_b32 %7, %7, %9, vcc"
          : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
          : "v"(a), "v"(b)
          : "vcc");
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t s = c;
#pragma unroll
  for (int i = 0; i < 8; i++) s ^= (uint32_t)r[i] ^ (uint32_t)(r[i] >> 32) ^ x[i];
  if (s == 0x12345u) sink[0] = s;
  if ((threadIdx.x & 63) == 0) {
    const int wave = t >> 6;
    tstamps[2 * wave] = t0;
    tstamps[2 * wave + 1] = t1;
  }
}

struct Row { const char* name; int kind; int insts_per_iter; };

template <int KIND>
static int run(const Row& row, uint64_t* d_ts, uint32_t* d_sink) {
  for (int W : {1, 2, 4, 8}) {
    const int blocks = 256 * W;
    const int waves = blocks * 4;
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d_ts, d_sink, 1u);  // warm-up
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d_ts, d_sink, 2u);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> ts(2 * waves);
    CK(hipMemcpy(ts.data(), d_ts, ts.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> el(waves);
    for (int i = 0; i < waves; i++) el[i] = (double)(ts[2 * i + 1] - ts[2 * i]);
    std::sort(el.begin(), el.end());
    const double med = el[waves / 2];
    const double n = (double)ITERS * row.insts_per_iter;
    // s_memtime ticks at a fixed 100 MHz on gfx9; also report by wall time at the nominal 2.4 GHz
    printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"memtime_ticks_median\": %.0f, \"kernel_ms\": %.4f, "
           "\"ns_per_wave_inst_per_simd\": %.4f, \"cycles_at_2.4GHz\": %.3f}\n",
           row.name, W, med, ms, ms * 1e6 / (n * W), ms * 1e6 / (n * W) * 2.4);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  }
  return 0;
}

int main() {
  uint64_t* d_ts;
  uint32_t* d_sink;
  CK(hipMalloc(&d_ts, 2 * 256 * 8 * 4 * 8));
  CK(hipMalloc(&d_sink, 64));
  static const Row rows[] = {
      {"v_mad_u64_u32 (independent)", 0, 16}, {"v_mad_u64_u32 (dependent chain)", 1, 16}, {"v_add_u32", 2, 16},
      {"v_add_co/v_addc_co chain x8", 3, 16}, {"v_mad_u64_u32 + v_addc_co (mac96)", 4, 16}, {"v_mul_lo_u32", 5, 16},
      {"v_mul_hi_u32", 6, 16}, {"v_lshrrev_b64", 7, 16}, {"v_lshl_add_u64", 8, 16}, {"v_and_b32", 9, 16},
      {"v_cndmask_b32", 10, 16}, {"v_add3_u32", 11, 16}, {"v_mad_u32_u24", 12, 16}, {"v_mad_i64_i32", 13, 16},
      {"v_alignbit_b32", 14, 16}, {"v_mov_b32 dpp quad_perm", 15, 16}, {"v_sub_u32 + v_add_u32", 16, 16},
      {"4 mad + and + lshr64 column", 17, 48}, {"v_mul_u32_u24", 18, 16}, {"v_sub_co/v_subb_co x8 + cndmask x8", 19, 16},
      {"v_fma_f64", 20, 16}, {"v_add_f64", 21, 16}};
  int rc = 0;
  rc |= run<0>(rows[0], d_ts, d_sink); rc |= run<1>(rows[1], d_ts, d_sink); rc |= run<2>(rows[2], d_ts, d_sink);
  rc |= run<3>(rows[3], d_ts, d_sink); rc |= run<4>(rows[4], d_ts, d_sink); rc |= run<5>(rows[5], d_ts, d_sink);
  rc |= run<6>(rows[6], d_ts, d_sink); rc |= run<7>(rows[7], d_ts, d_sink); rc |= run<8>(rows[8], d_ts, d_sink);
  rc |= run<9>(rows[9], d_ts, d_sink); rc |= run<10>(rows[10], d_ts, d_sink); rc |= run<11>(rows[11], d_ts, d_sink);
  rc |= run<12>(rows[12], d_ts, d_sink); rc |= run<13>(rows[13], d_ts, d_sink); rc |= run<14>(rows[14], d_ts, d_sink);
  rc |= run<15>(rows[15], d_ts, d_sink); rc |= run<16>(rows[16], d_ts, d_sink); rc |= run<17>(rows[17], d_ts, d_sink);
  rc |= run<18>(rows[18], d_ts, d_sink); rc |= run<19>(rows[19], d_ts, d_sink);
  rc |= run<20>(rows[20], d_ts, d_sink); rc |= run<21>(rows[21], d_ts, d_sink);
  return rc;
}
