// Instruction-rate and field-multiply micro-benchmarks for gfx950.  Measurement tooling for
// DESIGN.md's VALU roofline ("measure the real v_mad_u64_u32 issue rate first");
// not on the product path.
#include "curves.hpp"
#include "fp29.hpp"

namespace ncg {

enum UbKind : int {
  UB_MAD_U64_U32 = 0,
  UB_MUL_LO_U32 = 1,
  UB_MUL_HI_U32 = 2,
  UB_MAD_U32_U24 = 3,
  UB_ADDC_U32 = 4,
  UB_ADD_U64 = 5,
  UB_FMA_F64 = 6,
  UB_FMA_F32 = 7,
  UB_MODMUL_SECP = 8,
  UB_MODMUL_BLS = 9,
  UB_MODSQR_BLS = 10,
  UB_MODADD_BLS = 11,
  UB_MUL_HI_U24 = 12,
  UB_MODMUL_BLS29 = 13,
  UB_MODSQR_BLS29 = 14,
};

__device__ __forceinline__ uint32_t __umul24hi_sub(uint32_t x, uint32_t y) {
  uint32_t r;
  asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

template <int KIND>
__global__ void __launch_bounds__(256) k_ub_instr(uint32_t* out, const uint32_t* in, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a = in[t & 1023] | 1u, b = in[(t + 7) & 1023] | 3u;
  constexpr int CH = 8;  // independent chains per lane
  if constexpr (KIND == UB_MAD_U64_U32) {
    uint64_t acc[CH];
    for (int c = 0; c < CH; c++) acc[c] = a + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = (uint64_t)(uint32_t)acc[c] * b + acc[c];
    }
    uint64_t s = 0;
    for (int c = 0; c < CH; c++) s ^= acc[c];
    out[t] = (uint32_t)s ^ (uint32_t)(s >> 32);
  } else if constexpr (KIND == UB_MUL_LO_U32 || KIND == UB_MUL_HI_U32 || KIND == UB_MAD_U32_U24 ||
                       KIND == UB_MUL_HI_U24) {
    uint32_t acc[CH];
    for (int c = 0; c < CH; c++) acc[c] = a + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        if constexpr (KIND == UB_MUL_LO_U32) acc[c] = acc[c] * b;
        if constexpr (KIND == UB_MUL_HI_U32) acc[c] = __umulhi(acc[c], b) + 0x9e3779b9u;
        if constexpr (KIND == UB_MAD_U32_U24) acc[c] = __umul24(acc[c], b) + a;
        if constexpr (KIND == UB_MUL_HI_U24) acc[c] = __umul24hi_sub(acc[c], b) ^ a;
      }
    }
    uint32_t s = 0;
    for (int c = 0; c < CH; c++) s ^= acc[c];
    out[t] = s;
  } else if constexpr (KIND == UB_ADDC_U32) {
    uint32_t acc[CH];
    for (int c = 0; c < CH; c++) acc[c] = a + c;
    uint32_t cy = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = __builtin_addc(acc[c], b, cy, &cy);
    }
    uint32_t s = cy;
    for (int c = 0; c < CH; c++) s ^= acc[c];
    out[t] = s;
  } else if constexpr (KIND == UB_ADD_U64) {
    uint64_t acc[CH];
    uint64_t bb = ((uint64_t)b << 32) | a;
    for (int c = 0; c < CH; c++) acc[c] = a + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = acc[c] + bb + (acc[c] >> 63);
    }
    uint64_t s = 0;
    for (int c = 0; c < CH; c++) s ^= acc[c];
    out[t] = (uint32_t)s ^ (uint32_t)(s >> 32);
  } else if constexpr (KIND == UB_FMA_F64) {
    double acc[CH];
    double x = 1.0 + (double)(a & 0xff) * 1e-9, y = (double)(b & 0xff) * 1e-9;
    for (int c = 0; c < CH; c++) acc[c] = (double)c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = __builtin_fma(acc[c], x, y);
    }
    double s = 0;
    for (int c = 0; c < CH; c++) s += acc[c];
    out[t] = (uint32_t)(long long)s;
  } else if constexpr (KIND == UB_FMA_F32) {
    float acc[CH];
    float x = 1.0f + (float)(a & 0xff) * 1e-6f, y = (float)(b & 0xff) * 1e-6f;
    for (int c = 0; c < CH; c++) acc[c] = (float)c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = __builtin_fmaf(acc[c], x, y);
    }
    float s = 0;
    for (int c = 0; c < CH; c++) s += acc[c];
    out[t] = (uint32_t)(int)s;
  }
}

template <class PR, int OP>  // OP 0: mul, 1: sqr, 2: add
__global__ void __launch_bounds__(256) k_ub_field(uint32_t* out, const uint32_t* in, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fp<PR> a, b;
  for (int i = 0; i < PR::N; i++) {
    a.v[i] = in[(t + i) & 1023];
    b.v[i] = in[(t + 31 * i + 5) & 1023];
  }
  a.v[PR::N - 1] &= 0x0fffffffu;
  b.v[PR::N - 1] &= 0x0fffffffu;
  for (int i = 0; i < iters; i++) {
    if constexpr (OP == 0) a = fp_mul<PR>(a, b);
    if constexpr (OP == 1) a = fp_sqr<PR>(a);
    if constexpr (OP == 2) a = a + b;
  }
  uint32_t s = 0;
  for (int i = 0; i < PR::N; i++) s ^= a.v[i];
  out[t] = s;
}

template <int OP>
struct UbMul29 {
  static __device__ __noinline__ void run(uint32_t (&r)[14], const uint32_t (&a)[14], const uint32_t (&b)[14]) {
    if constexpr (OP == 0) mont_mul29<ParamsBls29>(r, a, b);
    else mont_sqr29<ParamsBls29>(r, a);
  }
};
template <int OP>
__global__ void __launch_bounds__(256) k_ub_field29(uint32_t* out, const uint32_t* in, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a[14], b[14];
  for (int i = 0; i < 14; i++) {
    a[i] = in[(t + i) & 1023] & 0x1fffffffu;
    b[i] = in[(t + 31 * i + 5) & 1023] & 0x1fffffffu;
  }
  a[13] &= 0xffffu;
  b[13] &= 0xffffu;
  for (int i = 0; i < iters; i++) UbMul29<OP>::run(a, a, b);
  uint32_t s = 0;
  for (int i = 0; i < 14; i++) s ^= a[i];
  out[t] = s;
}

// ---- field-level self-check on the device: out[i] = op(a[i], b[i]) for the field code the kernels use,
// one lane per element (tests/test_gpu_field.py compares with big-int arithmetic, so the inline-asm
// multiply-add chains and the bound-typed lazy reduction are pinned directly, not only through point
// operations).  field 0 / 1: fe9.hpp (secp256k1 / ed25519 p), operands as 9 RAW limbs each (the tests feed
// limbs at the top of what a bound type admits), `variant` = 10 A + B picks the operand bound types;
// field 2: bls12-381 Fe29 (canonical 12-word operands).  Results: canonical wire words (8 or 12) per element.
template <class PR, int A, int B>
__device__ void field_check_fe9(int op, const uint32_t* a, const uint32_t* b, uint32_t* r) {
  Fe9<PR, A> x;
  Fe9<PR, B> y;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    x.v[i] = a[i];
    y.v[i] = b[i];
  }
  switch (op) {
    case 0: fe9_to_wire(r, x * y); break;
    case 1: fe9_to_wire(r, f_sqr(x)); break;
    case 2: if constexpr (A + B <= 7) fe9_to_wire(r, x + y); break;
    case 3: if constexpr (A + B + 1 <= 7) fe9_to_wire(r, x - y); break;
    case 4: if constexpr (A + 1 <= 7) fe9_to_wire(r, f_neg(x)); break;
    case 5: fe9_to_wire(r, f_inv(x)); break;
    case 6: fe9_to_wire(r, fe9_norm(x)); break;
    case 7: {
      for (int i = 0; i < 8; i++) r[i] = 0;
      r[0] = f_eqz(x) ? 1u : 0u;
      break;
    }
    case 9: {  // largest limb of the raw product (output-bound check)
      auto n = x * y;
      uint32_t mx = 0;
      for (int i = 0; i < 9; i++) mx = n.v[i] > mx ? n.v[i] : mx;
      for (int i = 0; i < 8; i++) r[i] = 0;
      r[0] = mx;
      break;
    }
  }
}
template <class PR>
__global__ void __launch_bounds__(64) k_field_check_fe9(int op, int variant, const uint32_t* __restrict__ a,
                                                        const uint32_t* __restrict__ b, uint32_t* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t *pa = a + (size_t)i * 9, *pb = b + (size_t)i * 9;
  uint32_t* r = out + (size_t)i * 8;
  switch (variant) {
    case 11: field_check_fe9<PR, 1, 1>(op, pa, pb, r); break;
    case 12: field_check_fe9<PR, 1, 2>(op, pa, pb, r); break;
    case 17: field_check_fe9<PR, 1, 7>(op, pa, pb, r); break;
    case 71: field_check_fe9<PR, 7, 1>(op, pa, pb, r); break;
    case 23: field_check_fe9<PR, 2, 3>(op, pa, pb, r); break;
    case 22: field_check_fe9<PR, 2, 2>(op, pa, pb, r); break;
    case 33: field_check_fe9<PR, 3, 3>(op, pa, pb, r); break;
    case 77: field_check_fe9<PR, 7, 7>(op, pa, pb, r); break;
    default: break;
  }
}
__global__ void __launch_bounds__(64) k_field_check_fe29(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                         uint32_t* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fe29<2> x = fe29_from_wire(a + (size_t)i * 12), y = fe29_from_wire(b + (size_t)i * 12);
  uint32_t* r = out + (size_t)i * 12;
  switch (op) {
    case 0: fe29_to_wire(r, x * y); break;
    case 1: fe29_to_wire(r, f_sqr(x)); break;
    case 2: fe29_to_wire(r, x + y); break;
    case 3: fe29_to_wire(r, x - y); break;
    case 4: fe29_to_wire(r, f_neg(x)); break;
    case 5: fe29_to_wire(r, f_inv(x)); break;
    default: break;
  }
}
// Fe29 / lane-paired Fp2 on RAW limb arrays at the TOP of their value bounds (the column bounds of the fused products -
// "56 product + 14 reduction terms below 2^64 only because limb 13 of a value below 2^12 p is at most 53256", fe29.hpp -
// are only exercised by operands that generic values never reach).  Inputs per item: a = [a, c], b = [b, d] (each
// element 14 words unpaired / 28 words paired: c0 then c1); ops: 0 a*b, 1 a^2, 6 a*b - c*d; output: canonical wire.
//   unpaired (field 3): a, c < 4096 p;  b, d < 4096 p (mul) / 2048 p (mulsub)
//   paired   (field 4): a, c < 4096 p;  b, d < 2048 p (mul) / 1024 p (mulsub);  a < 2048 p for the square
template <int B>
__device__ __forceinline__ Fe29<B> fe29_raw(const uint32_t* p) {
  Fe29<B> r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = p[i];
  return r;
}
__global__ void __launch_bounds__(64) k_field_check_fe29raw(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                            uint32_t* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t *pa = a + (size_t)i * 28, *pb = b + (size_t)i * 28;
  uint32_t* r = out + (size_t)i * 12;
  switch (op) {
    case 0: fe29_to_wire(r, fe29_raw<4096>(pa) * fe29_raw<4096>(pb)); break;
    case 1: fe29_to_wire(r, f_sqr(fe29_raw<4096>(pa))); break;
    case 6: fe29_to_wire(r, f_mulsub(fe29_raw<4096>(pa), fe29_raw<2048>(pb), fe29_raw<4096>(pa + 14), fe29_raw<2048>(pb + 14))); break;
    default: break;
  }
}
__global__ void __launch_bounds__(64) k_field_check_fe29x2p(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                            uint32_t* __restrict__ out, int n) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;  // one Fp2 element per lane pair
  if (i >= n) return;
  const int h = pair_odd() ? 14 : 0;
  const uint32_t *pa = a + (size_t)i * 56 + h, *pb = b + (size_t)i * 56 + h;
  uint32_t* r = out + (size_t)i * 24 + (pair_odd() ? 12 : 0);
  switch (op) {
    case 0: fe29_to_wire(r, (Fe29x2P<4096>(fe29_raw<4096>(pa)) * Fe29x2P<2048>(fe29_raw<2048>(pb))).h); break;
    case 1: fe29_to_wire(r, f_sqr(Fe29x2P<2048>(fe29_raw<2048>(pa))).h); break;
    case 6:
      fe29_to_wire(r, f_mulsub(Fe29x2P<4096>(fe29_raw<4096>(pa)), Fe29x2P<1024>(fe29_raw<1024>(pb)),
                               Fe29x2P<4096>(fe29_raw<4096>(pa + 28)), Fe29x2P<1024>(fe29_raw<1024>(pb + 28))).h);
      break;
    default: break;
  }
}
hipError_t field_check_run(int field, int op, int variant, const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_out, int n,
                           hipStream_t st) {
  if (n <= 0) return hipSuccess;
  const dim3 grid((n + 63) / 64), block(64);
  if (field == 0) hipLaunchKernelGGL(k_field_check_fe9<Fe9SecpPR>, grid, block, 0, st, op, variant, d_a, d_b, d_out, n);
  else if (field == 1) hipLaunchKernelGGL(k_field_check_fe9<Fe9EdPR>, grid, block, 0, st, op, variant, d_a, d_b, d_out, n);
  else if (field == 2) hipLaunchKernelGGL(k_field_check_fe29, grid, block, 0, st, op, d_a, d_b, d_out, n);
  else if (field == 3) hipLaunchKernelGGL(k_field_check_fe29raw, grid, block, 0, st, op, d_a, d_b, d_out, n);
  else if (field == 4) hipLaunchKernelGGL(k_field_check_fe29x2p, dim3((2 * n + 63) / 64), block, 0, st, op, d_a, d_b, d_out, n);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// Returns milliseconds for one launch of `kind` with the given geometry (after one warm-up).
hipError_t ubench_run(int kind, int blocks, int threads, int iters, uint32_t* d_out, const uint32_t* d_in,
                      hipStream_t st, float* ms) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    if (rep == 1) (void)hipEventRecord(e0, st);
#define UB_LAUNCH(K) hipLaunchKernelGGL((K), dim3(blocks), dim3(threads), 0, st, d_out, d_in, iters)
    switch (kind) {
      case UB_MAD_U64_U32: UB_LAUNCH(k_ub_instr<UB_MAD_U64_U32>); break;
      case UB_MUL_LO_U32: UB_LAUNCH(k_ub_instr<UB_MUL_LO_U32>); break;
      case UB_MUL_HI_U32: UB_LAUNCH(k_ub_instr<UB_MUL_HI_U32>); break;
      case UB_MAD_U32_U24: UB_LAUNCH(k_ub_instr<UB_MAD_U32_U24>); break;
      case UB_MUL_HI_U24: UB_LAUNCH(k_ub_instr<UB_MUL_HI_U24>); break;
      case UB_ADDC_U32: UB_LAUNCH(k_ub_instr<UB_ADDC_U32>); break;
      case UB_ADD_U64: UB_LAUNCH(k_ub_instr<UB_ADD_U64>); break;
      case UB_FMA_F64: UB_LAUNCH(k_ub_instr<UB_FMA_F64>); break;
      case UB_FMA_F32: UB_LAUNCH(k_ub_instr<UB_FMA_F32>); break;
      case UB_MODMUL_SECP: UB_LAUNCH((k_ub_field<ParamsSecpP, 0>)); break;
      case UB_MODMUL_BLS: UB_LAUNCH((k_ub_field<ParamsBlsP, 0>)); break;
      case UB_MODSQR_BLS: UB_LAUNCH((k_ub_field<ParamsBlsP, 1>)); break;
      case UB_MODADD_BLS: UB_LAUNCH((k_ub_field<ParamsBlsP, 2>)); break;
      case UB_MODMUL_BLS29: UB_LAUNCH((k_ub_field29<0>)); break;
      case UB_MODSQR_BLS29: UB_LAUNCH((k_ub_field29<1>)); break;
      default: return hipErrorInvalidValue;
    }
#undef UB_LAUNCH
  }
  (void)hipEventRecord(e1, st);
  hipError_t err = hipEventSynchronize(e1);
  if (err == hipSuccess) err = hipEventElapsedTime(ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return err;
}

}  // namespace ncg
