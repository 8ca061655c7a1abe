// Batch fixed-base scalar multiplication: out[i] = k[i] * BASE.
//
// Replaces, batch-wise, BASE.multiply(k) / BASE.multiplyUnsafe(k) of the reference, which walk
// a cached table of window multiples with one addition per window and no doublings
// (ScalarMultiplier.buildWnafTable / wnafCachedCT, src/abstract/curve.ts:560-606; W = 6 set at
// src/abstract/weierstrass.ts:1018).  Same structure here, sized for the GPU: signed-odd digits
// of 8 bits (33 windows cover 257 bits + the carry), table[w][j] = (2j+1) * 2^(8w) * BASE as
// affine points in storage format - 33 * 128 entries (270 KB secp256k1, 473 KB G1, 946 KB G2),
// resident in the 4 MB L2 of every XCD - so one multiplication is 33 mixed additions, followed
// by the shared batched inversion.  The table is built once per context with the variable-base
// kernel itself.
#include <vector>

#include "host_api.hpp"
#include "mulvar.hpp"

namespace ncg {

constexpr int MB_W = 8, MB_M = 33, MB_T = 1 << (MB_W - 1);  // 33 windows x 128 entries

template <class C>
__global__ void __launch_bounds__(256) k_mul_base(const uint32_t* __restrict__ table,
                                                  const uint32_t* __restrict__ scalars, uint32_t* __restrict__ jac_out,
                                                  int n) {
  using F = typename C::F;
  constexpr int FW = FieldIO<F>::WORDS;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> LaneShift<C>::value;
  if (i >= n) return;
  uint32_t k[8];
#pragma unroll
  for (int j = 0; j < 8; j++) k[j] = scalars[(size_t)i * 8 + j];
  const bool zero = mp_is_zero<8>(k);
  SignedOddWindows<9, MB_W, MB_M> win;
  win.template init<8>(k);
  Jac<F> R = Jac<F>::inf();
  for (int w = MB_M - 1; w >= 0; w--) {
    int d = win.pop();
    const uint32_t* e = table + ((size_t)w * MB_T + (((d < 0 ? -d : d) - 1) >> 1)) * 2 * FW;
    Affine<F> q{FieldIO<F>::load(e), FieldIO<F>::load(e + FW)};
    if (d < 0) q.y = f_neg(q.y);
    R = jac_madd(R, q);
  }
  if (win.was_even) {  // the scalar was bumped by one: take BASE back out
    Affine<F> q{FieldIO<F>::load(table), FieldIO<F>::load(table + FW)};
    q.y = f_neg(q.y);
    R = jac_madd(R, q);
  }
  if (zero) R = Jac<F>::inf();
  uint32_t* o = jac_out + (size_t)i * 3 * FW;
  FieldIO<F>::store(o, R.X);
  FieldIO<F>::store(o + FW, R.Y);
  FieldIO<F>::store(o + 2 * FW, R.Z);
}

// affine wire points -> storage format (Montgomery limbs), used for the table
template <class C>
__global__ void __launch_bounds__(256) k_wire_to_storage(const uint32_t* __restrict__ wire, uint32_t* __restrict__ out,
                                                         int n) {
  using F = typename C::F;
  constexpr int FW = FieldIO<F>::WORDS, WW = FieldWire<F>::WORDS;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> LaneShift<C>::value;
  if (i >= n) return;
  Affine<F> a = load_affine_wire<F>(wire + (size_t)i * 2 * WW);
  FieldIO<F>::store(out + (size_t)i * 2 * FW, a.x);
  FieldIO<F>::store(out + (size_t)i * 2 * FW + FW, a.y);
}

template <class C>
static size_t table_bytes() {
  return (size_t)MB_M * MB_T * 2 * FieldIO<typename C::F>::WORDS * 4;
}

// Builds table[w][j] = (2j+1) * 2^(8w) * BASE on the device.  base_wire: affine wire BASE (host).
// order8: group order (the window base 2^(8w) is reduced mod the order: BASE has prime order).
template <class C>
static hipError_t build_table_t(int curve, const uint32_t* base_wire, uint32_t* d_table, hipStream_t st) {
  using F = typename C::F;
  constexpr int WW = FieldWire<F>::WORDS;
  const int NB = MB_M, NT = MB_M * MB_T;
  // scalars 2^(8w) for w < 32 fit in 256 bits; 2^256 = 2^256 - n (mod n) for every order on the path
  std::vector<uint32_t> h_pts((size_t)NT * 2 * WW), h_sc((size_t)NT * 8, 0);
  const uint32_t* order = curve == CURVE_SECP256K1 ? Orders::SECP_N : Orders::BLS_R;
  for (int w = 0; w < NB; w++) {
    for (int j = 0; j < 2 * WW; j++) h_pts[(size_t)w * 2 * WW + j] = base_wire[j];
    uint32_t* s = &h_sc[(size_t)w * 8];
    if (w < 32) {
      s[(8 * w) / 32] = 1u << ((8 * w) % 32);
    } else {
      // 2^256 mod order = 2^256 - q*order with q = floor(2^256 / order): 1 for secp256k1's n, 2 for
      // bls12-381's r; computed as the two's complement of q*order in 256 bits
      const int q = curve == CURVE_SECP256K1 ? 1 : 2;
      uint64_t cy = 0, bw = 0;
      for (int i = 0; i < 8; i++) {
        cy += (uint64_t)order[i] * q;
        uint64_t d = (uint64_t)0 - (uint32_t)cy - bw;
        s[i] = (uint32_t)d;
        bw = (d >> 32) & 1;
        cy >>= 32;
      }
    }
  }
  struct DevBuf {  // frees on every exit path
    void* p = nullptr;
    ~DevBuf() {
      if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
  } b_pts, b_sc, b_out, b_inf, b_jac;
  size_t pts_b = (size_t)NT * 2 * WW * 4, sc_b = (size_t)NT * 32;
  hipError_t e;
  if ((e = b_pts.alloc(pts_b)) != hipSuccess) return e;
  if ((e = b_sc.alloc(sc_b)) != hipSuccess) return e;
  if ((e = b_out.alloc(pts_b)) != hipSuccess) return e;
  if ((e = b_inf.alloc(NT)) != hipSuccess) return e;
  if ((e = b_jac.alloc(mul_var_tmp_bytes(curve, NT))) != hipSuccess) return e;
  uint32_t *d_pts = (uint32_t*)b_pts.p, *d_sc = (uint32_t*)b_sc.p, *d_out = (uint32_t*)b_out.p,
           *d_jac = (uint32_t*)b_jac.p;
  uint8_t* d_inf = (uint8_t*)b_inf.p;
  // step 1: B_w = 2^(8w) * BASE
  (void)hipMemcpyAsync(d_pts, h_pts.data(), (size_t)NB * 2 * WW * 4, hipMemcpyHostToDevice, st);
  (void)hipMemcpyAsync(d_sc, h_sc.data(), (size_t)NB * 32, hipMemcpyHostToDevice, st);
  e = mul_var_batch(curve, d_pts, d_sc, d_out, d_inf, NB, d_jac, st);
  if (e != hipSuccess) return e;
  std::vector<uint32_t> h_bw((size_t)NB * 2 * WW);
  (void)hipMemcpyAsync(h_bw.data(), d_out, h_bw.size() * 4, hipMemcpyDeviceToHost, st);
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  // step 2: table[w][j] = (2j+1) * B_w
  for (int w = 0; w < NB; w++)
    for (int j = 0; j < MB_T; j++) {
      size_t idx = (size_t)w * MB_T + j;
      for (int t = 0; t < 2 * WW; t++) h_pts[idx * 2 * WW + t] = h_bw[(size_t)w * 2 * WW + t];
      for (int t = 0; t < 8; t++) h_sc[idx * 8 + t] = 0;
      h_sc[idx * 8] = 2u * j + 1u;
    }
  (void)hipMemcpyAsync(d_pts, h_pts.data(), pts_b, hipMemcpyHostToDevice, st);
  (void)hipMemcpyAsync(d_sc, h_sc.data(), sc_b, hipMemcpyHostToDevice, st);
  e = mul_var_batch(curve, d_pts, d_sc, d_out, d_inf, NT, d_jac, st);
  if (e != hipSuccess) return e;
  using D = typename DeviceCurve<C>::type;
  hipLaunchKernelGGL(k_wire_to_storage<D>, dim3(((NT << LaneShift<D>::value) + 255) / 256), dim3(256), 0, st, d_out,
                     d_table, NT);
  return hipStreamSynchronize(st);
}

size_t mul_base_table_bytes(int curve) {
  switch (curve) {
    case CURVE_SECP256K1: return table_bytes<CurveSecp>();
    case CURVE_BLS12_381_G1: return table_bytes<CurveG1>();
    case CURVE_BLS12_381_G2: return table_bytes<CurveG2>();
    default: return 0;
  }
}

hipError_t mul_base_build_table(int curve, const uint32_t* base_wire, uint32_t* d_table, hipStream_t st) {
  switch (curve) {
    case CURVE_SECP256K1: return build_table_t<CurveSecp>(curve, base_wire, d_table, st);
    case CURVE_BLS12_381_G1: return build_table_t<CurveG1>(curve, base_wire, d_table, st);
    case CURVE_BLS12_381_G2: return build_table_t<CurveG2>(curve, base_wire, d_table, st);
    default: return hipErrorInvalidValue;
  }
}

template <class C, int K>
static hipError_t mul_base_t(const uint32_t* table, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                             uint32_t* jac_tmp, hipStream_t st) {
  constexpr int LS = LaneShift<C>::value;
  hipLaunchKernelGGL(k_mul_base<C>, dim3((unsigned)((((size_t)n << LS) + 255) / 256)), dim3(256), 0, st, table, scalars,
                     jac_tmp, n);
  int threads = ((n + K - 1) / K) << LS;
  hipLaunchKernelGGL((k_jac_batch_affine<C, K>), dim3((threads + 255) / 256), dim3(256), 0, st, jac_tmp, out, out_inf, n);
  return hipGetLastError();
}

hipError_t mul_base_batch(int curve, const uint32_t* table, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf,
                          int n, uint32_t* jac_tmp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  switch (curve) {
    case CURVE_SECP256K1: return mul_base_t<CurveSecp, 8>(table, scalars, out, out_inf, n, jac_tmp, st);
    case CURVE_BLS12_381_G1: return mul_base_t<CurveG1, 8>(table, scalars, out, out_inf, n, jac_tmp, st);
    case CURVE_BLS12_381_G2: return mul_base_t<CurveG2P, 4>(table, scalars, out, out_inf, n, jac_tmp, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ncg
