// secp256k1 ECDSA verification for a batch (the caller above Point.mulAddUnsafe):
//   verify(sig, msgHash, publicKey) of src/abstract/weierstrass.ts:1571-1620 with format 'compact' and
//   prehash: false - r, s in [1, n), optional low-S rule, h = bits2int_modN(msgHash), u1 = h s^-1, u2 = r s^-1
//   (mod n), R = u1 G + u2 P, accept iff R != O and R.x mod n == r.
// The group work is the existing batch pipeline (SEC1 decode -> fixed-base multiply -> variable-base multiply
// -> pairwise add); this file adds the scalar side: Montgomery arithmetic modulo the group order n (fp.hpp's
// product-scanning multiply with a parameter set for n), the s^-1 of K signatures per lane with one Fermat
// inversion (Montgomery's trick, the shape of FpInvertBatch, modular.ts:722-747), and the final comparison.
#include "host_api.hpp"
#include "scalar.hpp"
#include "sha256.hpp"

namespace ncg {

// Montgomery constants of the group order n (R = 2^256): python -c "n=0xFFFF...4141; R=1<<256; print(-pow(n,-1,2**32) % 2**32, R % n, R*R % n)"
struct ParamsSecpN {
  static constexpr int N = 8;
  static constexpr uint32_t P[8] = {0xd0364141u, 0xbfd25e8cu, 0xaf48a03bu, 0xbaaedce6u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  static constexpr uint32_t INV = 0x5588b13fu;  // -n^-1 mod 2^32
  static constexpr uint32_t R1[8] = {0x2fc9bebfu, 0x402da173u, 0x50b75fc4u, 0x45512319u, 0x00000001u, 0x00000000u, 0x00000000u, 0x00000000u};
  static constexpr uint32_t R2[8] = {0x67d7d140u, 0x896cf214u, 0x0e7cf878u, 0x741496c2u, 0x5bcd07c6u, 0xe697f5e4u, 0x81c69bc5u, 0x9d671cd5u};
  static constexpr bool TOP_SPARE = false;  // n > 2^255: the product keeps its carry word (fp_mul_fips_asm / fp_mul_body)
  static constexpr bool FOLD = false;
  static constexpr uint32_t HALF[8] = {0x681b20a0u, 0xdfe92f46u, 0x57a4501du, 0x5d576e73u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x7fffffffu};  // n >> 1
};
using Fn = Fp<ParamsSecpN>;

NCG_DI Fn fn_from_words(const uint32_t (&w)[8]) {
  Fn r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = w[i];
  return r;
}
// a^(n-2), a != 0 (Montgomery form in and out): 4-bit windows over the exponent, 252 squarings + <= 78 products
NCG_DI Fn fn_inv(const Fn& a) {
  Fn tab[16];
  tab[0] = Fn::one();
  tab[1] = a;
  for (int j = 2; j < 16; j++) tab[j] = tab[j - 1] * a;
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = ParamsSecpN::P[i];
  e[0] -= 2u;  // n - 2 (no borrow: the low limb is 0xd0364141)
  Fn acc = tab[e[7] >> 28];
  for (int w = 62; w >= 0; w--) {
    for (int s = 0; s < 4; s++) acc = fp_sqr<ParamsSecpN>(acc);
    const uint32_t dg = (e[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (dg) acc = acc * tab[dg];
  }
  return acc;
}

NCG_DI void be32_to_words(uint32_t (&w)[8], const uint8_t* __restrict__ p) {  // 32 big-endian bytes -> 8 LE limbs
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint8_t* q = p + 4 * (7 - i);
    w[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
  }
}
NCG_DI bool words_lt(const uint32_t (&a)[8], const uint32_t (&b)[8]) {
  uint32_t d[8];
  return mp_sub<8>(d, a, b) != 0;
}

// One lane: K consecutive signatures.  sig: r || s (32 big-endian bytes each, Signature.fromBytes 'compact',
// weierstrass.ts:1275-1290: both in [1, n)); hash: 32 bytes, h = bits2int_modN (:1333-1347: the integer of the
// bytes, reduced mod n).  Writes u1, u2 as 32-byte little-endian scalars (the wire format of the batch
// multiplies) and sig_ok = 0 where the reference returns false before any group operation.
template <int K>
NCG_DI void ecdsa_prepare_lane(const uint8_t* __restrict__ sig, const uint8_t* __restrict__ hash, int lo, int hi, bool low_s,
                               uint32_t* __restrict__ u1, uint32_t* __restrict__ u2, uint8_t* __restrict__ sig_ok) {
  uint32_t n8[8], half[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    n8[i] = ParamsSecpN::P[i];
    half[i] = ParamsSecpN::HALF[i];
  }
  const Fn r2 = Fn::from_const(ParamsSecpN::R2);
  Fn sv[K], pre[K];   // s_j and the running products s_0 ... s_{j-1}, Montgomery form
  bool ok[K];
  Fn acc = Fn::one();
  for (int j = 0; j < K; j++) {
    const int idx = lo + j;
    ok[j] = false;
    sv[j] = Fn::one();
    if (idx < hi) {
      uint32_t r[8], s[8];
      be32_to_words(r, sig + (size_t)idx * 64);
      be32_to_words(s, sig + (size_t)idx * 64 + 32);
      bool good = !mp_is_zero(r) && !mp_is_zero(s) && words_lt(r, n8) && words_lt(s, n8);
      if (low_s && good && words_lt(half, s)) good = false;  // hasHighS: s > n >> 1 (weierstrass.ts:1293-1295)
      ok[j] = good;
      if (good) sv[j] = fn_from_words(s) * r2;
    }
    pre[j] = acc;
    acc = acc * sv[j];
  }
  Fn inv = fn_inv(acc);
  for (int j = K - 1; j >= 0; j--) {
    const int idx = lo + j;
    const Fn is = inv * pre[j];  // s_j^-1 (Montgomery form)
    inv = inv * sv[j];
    if (idx >= hi) continue;
    uint32_t r[8], h[8];
    be32_to_words(r, sig + (size_t)idx * 64);
    be32_to_words(h, hash + (size_t)idx * 32);
    {  // h mod n: h < 2^256 < 2n
      uint32_t d8[8];
      const bool ge = mp_sub<8>(d8, h, n8) == 0;
#pragma unroll
      for (int i = 0; i < 8; i++) h[i] = ge ? d8[i] : h[i];
    }
    // plain * Montgomery -> plain: u1 = h s^-1, u2 = r s^-1 (r < n when ok; other rows are rejected anyway)
    const Fn a = fn_from_words(h) * is;
    uint32_t rr[8];
    {
      uint32_t d8[8];
      const bool ge = mp_sub<8>(d8, r, n8) == 0;
#pragma unroll
      for (int i = 0; i < 8; i++) rr[i] = ge ? d8[i] : r[i];
    }
    const Fn b = fn_from_words(rr) * is;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      u1[(size_t)idx * 8 + i] = ok[j] ? a.v[i] : 0u;
      u2[(size_t)idx * 8 + i] = ok[j] ? b.v[i] : 0u;
    }
    sig_ok[idx] = ok[j] ? 1 : 0;
  }
}

constexpr int ECDSA_K = 4;
__global__ void __launch_bounds__(64) k_ecdsa_prepare(const uint8_t* __restrict__ sig, const uint8_t* __restrict__ hash, int n,
                                                      int low_s, uint32_t* __restrict__ u1, uint32_t* __restrict__ u2,
                                                      uint8_t* __restrict__ sig_ok) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const long lo = (long)t * ECDSA_K;
  if (lo >= n) return;
  ecdsa_prepare_lane<ECDSA_K>(sig, hash, (int)lo, n, low_s != 0, u1, u2, sig_ok);
}

// accept iff the signature and the key were well-formed, R != O and R.x mod n == r  (weierstrass.ts:1607-1612)
__global__ void __launch_bounds__(256) k_ecdsa_finish(const uint8_t* __restrict__ sig, const uint32_t* __restrict__ R,
                                                      const uint8_t* __restrict__ R_inf, const uint8_t* __restrict__ sig_ok,
                                                      const uint8_t* __restrict__ pub_ok, const uint8_t* __restrict__ pub_inf, int n,
                                                      uint8_t* __restrict__ out_ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok = sig_ok[i] != 0 && pub_ok[i] != 0 && pub_inf[i] == 0 && R_inf[i] == 0;
  uint32_t x[8], n8[8], d8[8], r[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    x[j] = R[(size_t)i * 16 + j];
    n8[j] = ParamsSecpN::P[j];
  }
  const bool ge = mp_sub<8>(d8, x, n8) == 0;  // x < p < 2n
  be32_to_words(r, sig + (size_t)i * 64);
  uint32_t diff = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) diff |= (ge ? d8[j] : x[j]) ^ r[j];
  out_ok[i] = (ok && diff == 0) ? 1 : 0;
}

// ---- BIP-340 Schnorr verification (src/secp256k1.ts:228-258): R = s G + (n - e) P with P = lift_x(pk), accept iff
// R != O, y(R) even and x(R) == r.  e = challenge mod n arrives from the host shim (tagged SHA-256 of
// r || pk || m, 32 bytes big-endian); this kernel range-checks r in [1, p) and s in [1, n) and lays out the
// multiplier inputs: u1 = s, u2 = n - e (0 for e = 0), key = 02 || pk (lift_x picks the even root).
__global__ void __launch_bounds__(256) k_schnorr_prepare(const uint8_t* __restrict__ sig, const uint8_t* __restrict__ e32,
                                                         const uint8_t* __restrict__ pkx, int n, uint32_t* __restrict__ u1,
                                                         uint32_t* __restrict__ u2, uint8_t* __restrict__ pub33,
                                                         uint8_t* __restrict__ pre_ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t r[8], s[8], e[8], n8[8], p8[8], ne[8];
  be32_to_words(r, sig + (size_t)i * 64);
  be32_to_words(s, sig + (size_t)i * 64 + 32);
  be32_to_words(e, e32 + (size_t)i * 32);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    n8[j] = ParamsSecpN::P[j];
    p8[j] = ParamsSecpP::P[j];
  }
  const bool ok = !mp_is_zero(r) && words_lt(r, p8) && !mp_is_zero(s) && words_lt(s, n8) && words_lt(e, n8);
  mp_sub<8>(ne, n8, e);
  const bool ez = mp_is_zero(e);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u1[(size_t)i * 8 + j] = ok ? s[j] : 0u;
    u2[(size_t)i * 8 + j] = (ok && !ez) ? ne[j] : 0u;
  }
  pub33[(size_t)i * 33] = 2;
  for (int j = 0; j < 32; j++) pub33[(size_t)i * 33 + 1 + j] = pkx[(size_t)i * 32 + j];
  pre_ok[i] = ok ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_schnorr_finish(const uint8_t* __restrict__ sig, const uint32_t* __restrict__ R,
                                                        const uint8_t* __restrict__ R_inf, const uint8_t* __restrict__ pre_ok,
                                                        const uint8_t* __restrict__ pub_ok, const uint8_t* __restrict__ pub_inf, int n,
                                                        uint8_t* __restrict__ out_ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool ok = pre_ok[i] != 0 && pub_ok[i] != 0 && pub_inf[i] == 0 && R_inf[i] == 0;
  uint32_t r[8];
  be32_to_words(r, sig + (size_t)i * 64);
  uint32_t diff = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) diff |= R[(size_t)i * 16 + j] ^ r[j];
  const bool even_y = (R[(size_t)i * 16 + 8] & 1u) == 0;
  out_ok[i] = (ok && diff == 0 && even_y) ? 1 : 0;
}
hipError_t schnorr_prepare(const uint8_t* d_sig, const uint8_t* d_e, const uint8_t* d_pkx, int n, uint32_t* d_u1, uint32_t* d_u2,
                           uint8_t* d_pub33, uint8_t* d_pre_ok, hipStream_t st) {
  hipLaunchKernelGGL(k_schnorr_prepare, dim3((n + 255) / 256), dim3(256), 0, st, d_sig, d_e, d_pkx, n, d_u1, d_u2, d_pub33, d_pre_ok);
  return hipGetLastError();
}
hipError_t schnorr_finish(const uint8_t* d_sig, const uint32_t* d_R, const uint8_t* d_R_inf, const uint8_t* d_pre_ok,
                          const uint8_t* d_pub_ok, const uint8_t* d_pub_inf, int n, uint8_t* d_out_ok, hipStream_t st) {
  hipLaunchKernelGGL(k_schnorr_finish, dim3((n + 255) / 256), dim3(256), 0, st, d_sig, d_R, d_R_inf, d_pre_ok, d_pub_ok, d_pub_inf, n,
                     d_out_ok);
  return hipGetLastError();
}

// ---- public-key recovery (Signature.recoverPublicKey, weierstrass.ts:1391-1407): sig65 = recid || r || s.
// R = the point with x = r (+ n for recid 2, 3; must stay below p) and the parity of recid bit 0; u1 = -h r^-1,
// u2 = s r^-1 (mod n); Q = u1 G + u2 R, rejected if O.  One lane handles K signatures with one inversion of the
// product of their r values.  Writes the 33-byte encoding of R, u1, u2 and pre_ok.
template <int K>
NCG_DI void ecdsa_recover_lane(const uint8_t* __restrict__ sig65, const uint8_t* __restrict__ hash, int lo, int hi,
                               uint32_t* __restrict__ u1, uint32_t* __restrict__ u2, uint8_t* __restrict__ pub33,
                               uint8_t* __restrict__ pre_ok) {
  uint32_t n8[8], p8[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    n8[i] = ParamsSecpN::P[i];
    p8[i] = ParamsSecpP::P[i];
  }
  const Fn r2 = Fn::from_const(ParamsSecpN::R2);
  Fn rv[K], pre[K];
  bool ok[K];
  Fn acc = Fn::one();
  for (int j = 0; j < K; j++) {
    const int idx = lo + j;
    ok[j] = false;
    rv[j] = Fn::one();
    if (idx < hi) {
      const uint8_t* sg = sig65 + (size_t)idx * 65;
      const uint32_t recid = sg[0];
      uint32_t r[8], s[8], radj[8];
      be32_to_words(r, sg + 1);
      be32_to_words(s, sg + 33);
      bool good = recid < 4 && !mp_is_zero(r) && !mp_is_zero(s) && words_lt(r, n8) && words_lt(s, n8);
      const uint32_t cy = mp_add<8>(radj, r, n8);
      if (recid >= 2) good = good && cy == 0 && words_lt(radj, p8);  // Fp.isValid(r + n)
      ok[j] = good;
      if (good) rv[j] = fn_from_words(r) * r2;
      // R's encoding: prefix 02 for an even y (recid bit 0 clear), x = radj big-endian
      uint8_t* pk = pub33 + (size_t)idx * 33;
      pk[0] = (recid & 1u) ? 3 : 2;
#pragma unroll
      for (int w = 0; w < 8; w++) {
        const uint32_t v = recid >= 2 ? radj[7 - w] : r[7 - w];
        pk[1 + 4 * w] = (uint8_t)(v >> 24);
        pk[2 + 4 * w] = (uint8_t)(v >> 16);
        pk[3 + 4 * w] = (uint8_t)(v >> 8);
        pk[4 + 4 * w] = (uint8_t)v;
      }
    }
    pre[j] = acc;
    acc = acc * rv[j];
  }
  Fn inv = fn_inv(acc);
  for (int j = K - 1; j >= 0; j--) {
    const int idx = lo + j;
    const Fn ir = inv * pre[j];  // r_j^-1 (Montgomery form)
    inv = inv * rv[j];
    if (idx >= hi) continue;
    uint32_t s[8], h[8];
    be32_to_words(s, sig65 + (size_t)idx * 65 + 33);
    be32_to_words(h, hash + (size_t)idx * 32);
    {
      uint32_t d8[8];
      const bool ge = mp_sub<8>(d8, h, n8) == 0;
#pragma unroll
      for (int i = 0; i < 8; i++) h[i] = ge ? d8[i] : h[i];
    }
    uint32_t ss[8];
    {
      uint32_t d8[8];
      const bool ge = mp_sub<8>(d8, s, n8) == 0;
#pragma unroll
      for (int i = 0; i < 8; i++) ss[i] = ge ? d8[i] : s[i];
    }
    const Fn a = fn_from_words(h) * ir;   // h r^-1
    const Fn b = fn_from_words(ss) * ir;  // s r^-1
    uint32_t na[8];
    mp_sub<8>(na, n8, a.v);               // -(h r^-1): n - a, or 0 when a == 0
    const bool az = a.is_zero();
#pragma unroll
    for (int i = 0; i < 8; i++) {
      u1[(size_t)idx * 8 + i] = (ok[j] && !az) ? na[i] : 0u;
      u2[(size_t)idx * 8 + i] = ok[j] ? b.v[i] : 0u;
    }
    pre_ok[idx] = ok[j] ? 1 : 0;
  }
}
__global__ void __launch_bounds__(64) k_ecdsa_recover_prepare(const uint8_t* __restrict__ sig65, const uint8_t* __restrict__ hash, int n,
                                                              uint32_t* __restrict__ u1, uint32_t* __restrict__ u2,
                                                              uint8_t* __restrict__ pub33, uint8_t* __restrict__ pre_ok) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const long lo = (long)t * 4;
  if (lo >= n) return;
  ecdsa_recover_lane<4>(sig65, hash, (int)lo, n, u1, u2, pub33, pre_ok);
}
// ok = the signature was well-formed, R decoded and Q != O; rejected rows get a zeroed point
__global__ void __launch_bounds__(256) k_ecdsa_recover_finish(uint32_t* __restrict__ Q, const uint8_t* __restrict__ Q_inf,
                                                              const uint8_t* __restrict__ pre_ok, const uint8_t* __restrict__ pub_ok,
                                                              int n, uint8_t* __restrict__ out_ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool ok = pre_ok[i] != 0 && pub_ok[i] != 0 && Q_inf[i] == 0;
  if (!ok) {
#pragma unroll
    for (int j = 0; j < 16; j++) Q[(size_t)i * 16 + j] = 0;
  }
  out_ok[i] = ok ? 1 : 0;
}
hipError_t ecdsa_recover_prepare(const uint8_t* d_sig65, const uint8_t* d_hash, int n, uint32_t* d_u1, uint32_t* d_u2,
                                 uint8_t* d_pub33, uint8_t* d_pre_ok, hipStream_t st) {
  const int lanes = (n + 3) / 4;
  hipLaunchKernelGGL(k_ecdsa_recover_prepare, dim3((lanes + 63) / 64), dim3(64), 0, st, d_sig65, d_hash, n, d_u1, d_u2, d_pub33, d_pre_ok);
  return hipGetLastError();
}
hipError_t ecdsa_recover_finish(uint32_t* d_Q, const uint8_t* d_Q_inf, const uint8_t* d_pre_ok, const uint8_t* d_pub_ok, int n,
                                uint8_t* d_out_ok, hipStream_t st) {
  hipLaunchKernelGGL(k_ecdsa_recover_finish, dim3((n + 255) / 256), dim3(256), 0, st, d_Q, d_Q_inf, d_pre_ok, d_pub_ok, n, d_out_ok);
  return hipGetLastError();
}

// Uncompressed SEC1 keys (04 || x || y, 65 bytes): Point.fromBytes checks the prefix, 0 <= x, y < p and the curve
// equation (weierstrass.ts:589-597, isValidXY) - no square root.  out: affine wire (x, y), ok, inf = 0.
__global__ void __launch_bounds__(256) k_secp_load_uncompressed(const uint8_t* __restrict__ pub65, uint32_t* __restrict__ out,
                                                                uint8_t* __restrict__ ok_out, uint8_t* __restrict__ inf_out, int n) {
  using PR = ParamsSecpP;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* in = pub65 + (size_t)i * 65;
  uint32_t xw[8], yw[8], p8[8];
  be32_to_words(xw, in + 1);
  be32_to_words(yw, in + 33);
#pragma unroll
  for (int j = 0; j < 8; j++) p8[j] = PR::P[j];
  bool ok = in[0] == 4 && words_lt(xw, p8) && words_lt(yw, p8);
  FpSecp x, y, seven = FpSecp::zero();
#pragma unroll
  for (int j = 0; j < 8; j++) {
    x.v[j] = ok ? xw[j] : 0u;
    y.v[j] = ok ? yw[j] : 0u;
  }
  seven.v[0] = 7;
  ok = ok && (fp_sqr<PR>(y) == fp_sqr<PR>(x) * x + seven);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    out[(size_t)i * 16 + j] = ok ? xw[j] : 0u;
    out[(size_t)i * 16 + 8 + j] = ok ? yw[j] : 0u;
  }
  ok_out[i] = ok ? 1 : 0;
  inf_out[i] = 0;
}
hipError_t secp_load_uncompressed(const uint8_t* d_pub65, uint32_t* d_out, uint8_t* d_ok, uint8_t* d_inf, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_secp_load_uncompressed, dim3((n + 255) / 256), dim3(256), 0, st, d_pub65, d_out, d_ok, d_inf, n);
  return hipGetLastError();
}

// ---- message hashing on the device (sha256.hpp): one lane per item; msgs = all messages back to back,
// msg_off = n + 1 byte offsets.  mode 0: out[i] = SHA-256(msg_i) (the prehash of ecdsa.verify); mode 1: the BIP-340
// challenge e_i = int(SHA-256(tag || tag || r_i || pk_i || msg_i)) mod n as 32 big-endian bytes.
struct Bip340Tag {  // SHA-256("BIP0340/challenge")
  static constexpr uint8_t H[32] = {0x7b, 0xb5, 0x2d, 0x7a, 0x9f, 0xef, 0x58, 0x32, 0x3e, 0xb1, 0xbf, 0x7a, 0x40, 0x7d, 0xb3, 0x82,
                                    0xd2, 0xf3, 0xf2, 0xd8, 0x1b, 0xb1, 0x22, 0x4f, 0x49, 0xfe, 0x51, 0x8f, 0x6d, 0x48, 0xd3, 0x7c};
};
__global__ void __launch_bounds__(256) k_sha256_msgs(const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ msg_off,
                                                     const uint8_t* __restrict__ sig64, const uint8_t* __restrict__ pkx, int mode, int n,
                                                     uint8_t* __restrict__ out32) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t lo = msg_off[i], hi = msg_off[i + 1];
  uint32_t h[8];
  if (mode == 0) {
    sha256_3(h, nullptr, 0, nullptr, 0, msgs + lo, hi - lo);
  } else {
    uint8_t pre[128];
    for (int j = 0; j < 32; j++) {
      pre[j] = Bip340Tag::H[j];
      pre[32 + j] = Bip340Tag::H[j];
      pre[64 + j] = sig64[(size_t)i * 64 + j];
      pre[96 + j] = pkx[(size_t)i * 32 + j];
    }
    sha256_3(h, pre, 128, nullptr, 0, msgs + lo, hi - lo);
    // e = digest mod n: the digest is below 2^256 < 2n
    uint32_t v[8], n8[8], d8[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      v[j] = h[7 - j];
      n8[j] = ParamsSecpN::P[j];
    }
    const bool ge = mp_sub<8>(d8, v, n8) == 0;
#pragma unroll
    for (int j = 0; j < 8; j++) h[7 - j] = ge ? d8[j] : v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    out32[(size_t)i * 32 + 4 * j] = (uint8_t)(h[j] >> 24);
    out32[(size_t)i * 32 + 4 * j + 1] = (uint8_t)(h[j] >> 16);
    out32[(size_t)i * 32 + 4 * j + 2] = (uint8_t)(h[j] >> 8);
    out32[(size_t)i * 32 + 4 * j + 3] = (uint8_t)h[j];
  }
}
hipError_t sha256_msgs(const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sig64, const uint8_t* d_pkx, int mode, int n,
                       uint8_t* d_out32, hipStream_t st) {
  hipLaunchKernelGGL(k_sha256_msgs, dim3((n + 255) / 256), dim3(256), 0, st, d_msgs, d_off, d_sig64, d_pkx, mode, n, d_out32);
  return hipGetLastError();
}

hipError_t ecdsa_prepare(const uint8_t* d_sig, const uint8_t* d_hash, int n, bool low_s, uint32_t* d_u1, uint32_t* d_u2,
                         uint8_t* d_sig_ok, hipStream_t st) {
  const int lanes = (n + ECDSA_K - 1) / ECDSA_K;
  hipLaunchKernelGGL(k_ecdsa_prepare, dim3((lanes + 63) / 64), dim3(64), 0, st, d_sig, d_hash, n, low_s ? 1 : 0, d_u1, d_u2, d_sig_ok);
  return hipGetLastError();
}
hipError_t ecdsa_finish(const uint8_t* d_sig, const uint32_t* d_R, const uint8_t* d_R_inf, const uint8_t* d_sig_ok,
                        const uint8_t* d_pub_ok, const uint8_t* d_pub_inf, int n, uint8_t* d_out_ok, hipStream_t st) {
  hipLaunchKernelGGL(k_ecdsa_finish, dim3((n + 255) / 256), dim3(256), 0, st, d_sig, d_R, d_R_inf, d_sig_ok, d_pub_ok, d_pub_inf, n,
                     d_out_ok);
  return hipGetLastError();
}

// host entry for the CPU tests (tests/hosttest): one signature through the lane code, K = 1
void ecdsa_prepare_host(const uint8_t* sig, const uint8_t* hash, bool low_s, uint32_t* u1, uint32_t* u2, uint8_t* ok) {
  ecdsa_prepare_lane<1>(sig, hash, 0, 1, low_s, u1, u2, ok);
}

}  // namespace ncg
