// Internal C++ interface between the C ABI (api.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ncg {

// jac_tmp: device scratch of mul_var_tmp_bytes(curve, n) bytes, or nullptr (per-lane inversion)
hipError_t mul_var_batch(int curve, const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf,
                         int n, uint32_t* jac_tmp, hipStream_t st);
size_t mul_var_tmp_bytes(int curve, int n);
// A/B variant of the secp256k1 kernel with the field multiply inlined (mulvar_inl.hip); minw = waves/SIMD requested
hipError_t mul_var_secp_inline(int minw, const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                               uint32_t* jac_tmp, hipStream_t st);
// projective (X, Y, Z) wire points -> affine wire points, x = X/Z, y = Y/Z (normalizeZ)
hipError_t normalize_batch(int curve, const uint32_t* proj_wire, uint32_t* out_wire, uint8_t* out_inf, int n,
                           hipStream_t st);

// pairwise A[i] + B[i] (subtract: A[i] - B[i]) on affine wire points; jac_tmp as for mul_var_batch
hipError_t pair_add_batch(int curve, const uint32_t* a, const uint32_t* b, int subtract, uint32_t* out, uint8_t* out_inf,
                          int n, uint32_t* jac_tmp, hipStream_t st);
hipError_t ed25519_proj_to_affine(const uint32_t* proj, uint32_t* out, uint8_t* out_inf, int n, hipStream_t st);

// fixed-base batch multiply (mulbase.hip)
size_t mul_base_table_bytes(int curve);
hipError_t mul_base_build_table(int curve, const uint32_t* base_wire_host, uint32_t* d_table, hipStream_t st);
hipError_t mul_base_batch(int curve, const uint32_t* table, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf,
                          int n, uint32_t* jac_tmp, hipStream_t st);

// batch point decoding (decode.hip): bytes per encoded point, 0 if the curve has no decoder
int decode_in_bytes(int curve);
hipError_t decode_points_batch(int curve, const uint8_t* in, int flags, uint32_t* out, uint8_t* ok, uint8_t* inf, int n,
                               hipStream_t st);
// compressed encodings of affine wire points (Point.toBytes); ok = 0 where the reference throws
hipError_t encode_points_batch(int curve, const uint32_t* in, uint8_t* out, uint8_t* ok, int n, hipStream_t st);
void encode_points_host(int curve, const uint32_t* in, uint8_t* out, uint8_t* ok, int n);
void decode_points_host(int curve, const uint8_t* in, int flags, uint32_t* out, uint8_t* ok, uint8_t* inf, int n);

// batch map-to-curve + cofactor clearing for bls12-381 G1 / G2 (h2c.hip); count = 1 or 2 field
// elements per output point
hipError_t map_to_curve_batch(int curve, const uint32_t* u, int count, uint32_t* out, uint8_t* inf, int n,
                              uint32_t* jac_tmp, hipStream_t st);
void map_to_curve_host(int curve, const uint32_t* u, int count, uint32_t* out, uint8_t* inf, int n);

// radix-2 NTT over bls12-381 Fr (ntt.hip)
size_t ntt_table_bytes(int n);
int ntt_tw_words();
void ntt_tw_from_canonical(const uint32_t (&w)[8], uint32_t* out);
size_t ntt_small_bytes(int n);
hipError_t ntt_build_table(int n, const uint32_t* d_omega, uint32_t* d_small, uint32_t* d_tab, hipStream_t st);
hipError_t ntt_run(int n, size_t batch, const uint32_t* src, uint32_t* dst, uint32_t* ws, const uint32_t* tab,
                   int tab_log, int flags, hipStream_t st);
int ntt_host(int n, const uint32_t* omega_wire, const uint32_t* src, uint32_t* dst, int flags, int t0max = 10, int tmax = 8);

struct MsmPlan;
// Optional second stream of a context: the wire -> storage conversion of the points has no consumer before
// the accumulate kernel, so it runs beside the digit / sort kernels (LDS- and memory-bound) instead of in
// front of them.  fork / join are timing-disabled events.
struct MsmSide {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  // set by the host-pointer entry point: the stored points are being produced elsewhere (chunked upload + conversion on
  // other streams); the accumulate kernel waits for this event, the digit / sort kernels do not
  hipEvent_t pts_ready = nullptr;
};

int msm_make_plan(int curve, int n, int c_override, MsmPlan* pl);
size_t msm_workspace_bytes(int curve, const MsmPlan& pl);
// d_pts / d_scalars: device; out_*: host.  Synchronises `st` (host-side Horner finish).
// *bad_index (optional): smallest index of a scalar >= the group order, 0xFFFFFFFF if none (the result is
// then meaningless: the caller fails the call like the reference's validateMSMScalars, curve.ts:398-404)
hipError_t msm_run(int curve, const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws,
                   uint32_t* out_affine_host, uint8_t* out_inf_host, hipStream_t st, uint32_t* bad_index = nullptr,
                   const MsmSide* side = nullptr);
// wire points -> the accumulate kernel's storage format (what a plan with pts_stored = 1 takes as d_pts)
size_t msm_stored_words_per_point(int curve);
hipError_t msm_points_to_stored(int curve, const uint32_t* d_pts_wire, int n, uint32_t* d_out, hipStream_t st);

// The same in two phases, for the multi-GPU path: msm_device_phase leaves the grouped window sums
// (msm_fin_words(curve, pl) words = npoints accumulators of msm_acc_words(curve) words) in the workspace
// and returns their device address; msm_sum_partials adds nparts such arrays element by element;
// msm_finish brings one array to the host and finishes (synchronises `st`).
hipError_t msm_device_phase(int curve, const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws,
                            const uint32_t** d_fin, hipStream_t st, const uint32_t** d_bad = nullptr,
                            const MsmSide* side = nullptr);
hipError_t msm_finish(int curve, const MsmPlan& pl, const uint32_t* d_fin, uint32_t* out_affine_host,
                      uint8_t* out_inf_host, hipStream_t st, const uint32_t* d_bad = nullptr, uint32_t* bad_host = nullptr);
hipError_t msm_sum_partials(int curve, uint32_t* d_gathered, int nparts, size_t npoints, uint32_t* d_out,
                            hipStream_t st);  // d_gathered is scratch: reduced in place
// asynchronous form: device phase + D2H of the grouped sums and the scalar-range flag into `land` (pinned host memory,
// msm_fin_words + 1 words), nothing synchronised; msm_finish_host is the host half of msm_finish on a host array of
// [ngroups(c)][nwin] grouped sums (also what the window-sharded multi-GPU mode assembles from the ranks' slots)
hipError_t msm_enqueue(int curve, const MsmPlan& pl, const uint32_t* d_pts, const uint32_t* d_scalars, void* ws, uint32_t* land,
                       hipStream_t st, const MsmSide* side = nullptr);
void msm_finish_host(int curve, int c, int nwin, const uint32_t* fin_host, uint32_t* out_affine_host, uint8_t* out_inf_host);
// the host finish of this curve is about to run (tens of microseconds from now): wakes its helper threads, if it has any
// (bls12-381: bls_host64.hpp FinishPool); a no-op otherwise
void msm_finish_prewake(int curve);
size_t msm_fin_words(int curve, const MsmPlan& pl);
size_t msm_acc_words(int curve);

// Window-shifted copies of a stored affine set for the shared-bucket MSM (msm_precomp.hip): d_levels holds nlev
// levels of m points (level 0 = the set itself, filled by the caller); level w = 2^(c w) * level 0.  Weierstrass
// curves only.  A plan with shared = 1 (and c, nwin = nlev) then takes d_levels as its point array.
size_t msm_shift_tmp_bytes(int curve, int m);
hipError_t msm_shift_levels(int curve, uint32_t* d_levels, int m, int nlev, int c, void* d_tmp, hipStream_t st);

// G1 batch multiply on verified subgroup points (mulvar_endo.hip): GLV ladder with the phi endomorphism
hipError_t mul_var_batch_g1_subgroup(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                                     uint32_t* jac_tmp, hipStream_t st);
// G2 likewise: four 64-bit streams along psi; jac_tmp holds mul_var_g2_subgroup_tmp_bytes(n) bytes
hipError_t mul_var_batch_g2_subgroup(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                                     uint32_t* jac_tmp, hipStream_t st);
size_t mul_var_g2_subgroup_tmp_bytes(int n);

// secp256k1 ECDSA batch verify, scalar side (ecdsa.hip): sig = r || s (big-endian), hash = 32 bytes.
hipError_t ecdsa_prepare(const uint8_t* d_sig, const uint8_t* d_hash, int n, bool low_s, uint32_t* d_u1, uint32_t* d_u2,
                         uint8_t* d_sig_ok, hipStream_t st);
hipError_t ecdsa_finish(const uint8_t* d_sig, const uint32_t* d_R, const uint8_t* d_R_inf, const uint8_t* d_sig_ok,
                        const uint8_t* d_pub_ok, const uint8_t* d_pub_inf, int n, uint8_t* d_out_ok, hipStream_t st);
hipError_t sha256_msgs(const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sig64, const uint8_t* d_pkx, int mode, int n,
                       uint8_t* d_out32, hipStream_t st);
hipError_t ecdsa_recover_prepare(const uint8_t* d_sig65, const uint8_t* d_hash, int n, uint32_t* d_u1, uint32_t* d_u2,
                                 uint8_t* d_pub33, uint8_t* d_pre_ok, hipStream_t st);
hipError_t ecdsa_recover_finish(uint32_t* d_Q, const uint8_t* d_Q_inf, const uint8_t* d_pre_ok, const uint8_t* d_pub_ok, int n,
                                uint8_t* d_out_ok, hipStream_t st);
hipError_t secp_load_uncompressed(const uint8_t* d_pub65, uint32_t* d_out, uint8_t* d_ok, uint8_t* d_inf, int n, hipStream_t st);
hipError_t schnorr_prepare(const uint8_t* d_sig, const uint8_t* d_e, const uint8_t* d_pkx, int n, uint32_t* d_u1, uint32_t* d_u2,
                           uint8_t* d_pub33, uint8_t* d_pre_ok, hipStream_t st);
hipError_t schnorr_finish(const uint8_t* d_sig, const uint32_t* d_R, const uint8_t* d_R_inf, const uint8_t* d_pre_ok,
                          const uint8_t* d_pub_ok, const uint8_t* d_pub_inf, int n, uint8_t* d_out_ok, hipStream_t st);
void ecdsa_prepare_host(const uint8_t* sig, const uint8_t* hash, bool low_s, uint32_t* u1, uint32_t* u2, uint8_t* ok);

// Endomorphism mode for bls12-381 point sets verified to lie in the prime-order subgroup (msm_endo.hip,
// endo.hpp).  msm_endo_factor: sub-scalars per scalar (2 on G1, 4 on G2, 0 = not offered).  msm_endo_expand
// writes the factor * n images of n wire points in the accumulate kernel's input format; a plan from
// msm_make_plan_endo makes msm_device_phase / msm_run take that array as `d_pts`.  msm_endo_verify compares
// [z^2]P (G1) / [z]P (G2) from the generic batch multiply with image 1: *d_bad = smallest failing index.
int msm_endo_factor(int curve);
int msm_make_plan_endo(int curve, int n_src, int c_override, MsmPlan* pl);
size_t msm_endo_words_per_point(int curve);
hipError_t msm_endo_expand(int curve, const uint32_t* d_pts_wire, int n, uint32_t* d_out, hipStream_t st);
hipError_t msm_endo_digits(const MsmPlan& pl, const uint32_t* d_scalars, int16_t* digits, uint32_t* bad, hipStream_t st);
hipError_t msm_endo_verify(int curve, const uint32_t* d_mult, const uint8_t* d_mult_inf, const uint32_t* d_images, int n,
                           uint32_t* d_bad, hipStream_t st);
void msm_endo_verify_scalar(int curve, uint32_t (&k)[8]);

// ed25519 batch verify (ed25519.hip).  btab: device copy of the table built by ed25519_build_base_table.
constexpr int ED25519_BTAB_WORDS = 256 * 27;  // [1,3,..,255] B then [1,3,..,255] 2^128 B, affine Niels, 3 x 9 stored words each
void ed25519_build_base_table(uint32_t* out_words);
size_t ed25519_verify_tmp_words(int n);
size_t ed25519_tmp_words(int n);
hipError_t ed25519_verify_batch(const uint32_t* sigs, const uint32_t* pks, const uint32_t* ks, const uint32_t* btab,
                                int zip215, uint8_t* out_ok, int n, uint32_t* gtab, hipStream_t st);
// challenge scalars k[i] = SHA-512(R_i || A_i || M_i) mod L (edwards.ts:984, :900-906) for a batch; ks: n x 8 words
hipError_t ed25519_challenge_batch(const uint8_t* sigs, const uint8_t* pks, const uint8_t* msgs, const uint64_t* msg_off,
                                   uint32_t* ks, int n, hipStream_t st);
void ed25519_challenge_host(const uint8_t* sig, const uint8_t* pk, const uint8_t* msg, uint64_t len, uint32_t* k_out);
hipError_t ed25519_mul_var_batch(const uint32_t* pts, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                                 uint32_t* proj_tmp, hipStream_t st);
// fixed-base multiply: table of ed25519_fixed_table_words() words built by ed25519_build_fixed_table (host)
size_t ed25519_fixed_table_words();
void ed25519_build_fixed_table(uint32_t* out_words);
hipError_t ed25519_mul_base_batch(const uint32_t* table, const uint32_t* scalars, uint32_t* out, uint8_t* out_inf, int n,
                                  uint32_t* proj_tmp, hipStream_t st);
void ed25519_mul_var_host(const uint32_t* pt, const uint32_t* k, uint32_t* out, uint8_t* out_inf);
bool ed25519_verify_host(const uint32_t* sig, const uint32_t* pk, const uint32_t* k, const uint32_t* btab, bool zip215);

// field-level self-check (ubench.hip): out[i] = op(a[i], b[i]) on the device field code
hipError_t field_check_run(int field, int op, int variant, const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_out, int n,
                           hipStream_t st);
hipError_t ubench_run(int kind, int blocks, int threads, int iters, uint32_t* d_out, const uint32_t* d_in,
                      hipStream_t st, float* ms);

}  // namespace ncg
