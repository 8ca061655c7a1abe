// Calibration kernels for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950: each kernel
// moves an exactly known number of HBM bytes in one of the access patterns the product kernels
// use, so the counter-to-bytes factor can be measured per pattern (MI355X_MICROARCH.md, HBM:
// "calibrate on a known byte count in your own access pattern").  Measurement tooling; not on
// the product path.   Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/_build/pmc_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib_fetch -- tools/_build/pmc_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/calib_write -- tools/_build/pmc_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// 16 B per lane, fully coalesced, each byte read once
__global__ void __launch_bounds__(256) calib_stream_read16(const uint4* __restrict__ src, uint32_t* __restrict__ sink, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// item-major gather: every lane reads ONE contiguous 64 B entry (4 x 16 B) at a pseudo-random
// 64 B-aligned position - the window-table lookup pattern of k_mul_var_gtab
__global__ void __launch_bounds__(256) calib_gather64(const uint4* __restrict__ src, uint32_t* __restrict__ sink, size_t n64,
                                                      int per_lane) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t s = t * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  uint32_t acc = 0;
  for (int k = 0; k < per_lane; k++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const uint4* p = src + (s % n64) * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// item-major write: every lane writes its own contiguous 64 B entries (the table build pattern)
__global__ void __launch_bounds__(256) calib_write64(uint4* __restrict__ dst, size_t n64) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n64; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = make_uint4((uint32_t)i, 1, 2, 3);
    uint4* p = dst + i * 4;
    p[0] = v; p[1] = v; p[2] = v; p[3] = v;
  }
}
// 16 B per lane coalesced streaming write
__global__ void __launch_bounds__(256) calib_stream_write16(uint4* __restrict__ dst, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main() {
  const size_t BYTES = (size_t)2 << 30;  // 2 GiB: 8x the 256 MiB Infinity Cache
  uint4* buf;
  uint32_t* sink;
  CK(hipMalloc(&buf, BYTES));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, BYTES));
  const size_t n16 = BYTES / 16, n64 = BYTES / 64;
  const int blocks = 256 * 16;
  const int per_lane = 32;
  const size_t gather_lanes = n64 / per_lane;  // total gathered = BYTES
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(calib_stream_read16, dim3(blocks), dim3(256), 0, 0, buf, sink, n16);
    hipLaunchKernelGGL(calib_gather64, dim3((unsigned)(gather_lanes / 256)), dim3(256), 0, 0, buf, sink, n64, per_lane);
    hipLaunchKernelGGL(calib_write64, dim3(blocks), dim3(256), 0, 0, buf, n64);
    hipLaunchKernelGGL(calib_stream_write16, dim3(blocks), dim3(256), 0, 0, buf, n16);
  }
  CK(hipDeviceSynchronize());
  printf("{\"bytes_per_launch\": %zu}\n", BYTES);
  return 0;
}
