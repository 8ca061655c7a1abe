// tools/sustain_valu.hip : what a box SUSTAINS on pure vector arithmetic - no memory traffic, no LDS - over seconds, not the
// 0.1 ms of tools/valu_rates.hip: (a) independent v_mad_u64_u32 only, (b) the ladder's mix (one multiply-add : one plain add),
// each at 3 and 8 waves per SIMD, launched back to back for ~1.5 s.  Prints lane-operations per second.  Two boxes whose
// figures differ here differ in the clock they hold under load, whatever sysfs reports; boxes that agree here and differ on
// the secp256k1 ladder differ in the memory path.   hipcc --offload-arch=gfx950 -O3 tools/sustain_valu.hip -o tools/_build/sustain_valu
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
template <int MIX>
__global__ void __launch_bounds__(256) k_sustain(uint32_t* sink, uint32_t seed) {
  uint64_t r[8];
  uint32_t x[8];
  uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
#pragma unroll
  for (int i = 0; i < 8; i++) { r[i] = a + i; x[i] = b + i; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
      if (MIX) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += (uint32_t)r[i] + (uint32_t)(r[i] >> 32) + x[i];
  if (s == 0x12345678u) sink[0] = s;
}
template <int MIX>
static int run(const char* name, int waves_per_simd, uint32_t* d_sink, double secs) {
  const int blocks = 256 * waves_per_simd;   // 4 waves per block, 256 CUs x 4 SIMDs
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_sustain<MIX>, dim3(blocks), dim3(256), 0, 0, d_sink, 1u);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  int launches = 0;
  double el = 0;
  while (el < secs) {
    for (int i = 0; i < 8; i++) hipLaunchKernelGGL(k_sustain<MIX>, dim3(blocks), dim3(256), 0, 0, d_sink, 2u);
    launches += 8;
    CK(hipDeviceSynchronize());
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double lane_mads = (double)launches * blocks * 256 * ITERS * 8;
  printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"seconds\": %.3f, \"lane_mads_per_s\": %.4e%s}\n", name, waves_per_simd, el, lane_mads / el,
         MIX ? ", \"plain_per_mad\": 1" : "");
  return 0;
}
int main(int argc, char** argv) {
  uint32_t* d_sink;
  CK(hipMalloc(&d_sink, 64));
  int rc = 0;
  if (argc > 1 && (argv[1][0] == 'm')) {   // "mad8", "mad3", "mix8", "mix3": ONE point for 3 s (tools/clock_probe_cmd.sh reads the clock beside it)
    const int w = argv[1][3] - '0';
    return argv[1][1] == 'a' ? run<0>("v_mad_u64_u32", w, d_sink, 3.0) : run<1>("v_mad_u64_u32 + v_add_u32", w, d_sink, 3.0);
  }
  const bool all = argc > 1;   // any other argument: the whole occupancy curve
  for (int w : {1, 2, 3, 4, 6, 8}) {
    if (!all && w != 3 && w != 8) continue;
    rc |= run<0>("v_mad_u64_u32", w, d_sink, all ? 0.5 : 1.5);
  }
  for (int w : {1, 2, 3, 4, 6, 8}) {
    if (!all && w != 3 && w != 8) continue;
    rc |= run<1>("v_mad_u64_u32 + v_add_u32", w, d_sink, all ? 0.5 : 1.5);
  }
  return rc;
}
