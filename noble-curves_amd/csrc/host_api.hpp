// Internal C++ interface between the C ABI (api.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ncg {

hipError_t mul_var_batch(int curve, const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf,
                         int n, hipStream_t st);

hipError_t ubench_run(int kind, int blocks, int threads, int iters, uint32_t* d_out, const uint32_t* d_in,
                      hipStream_t st, float* ms);

}  // namespace ncg
