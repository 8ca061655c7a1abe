/* ncg.h - C ABI of the MI355X batch elliptic-curve engine ("noble-curves GPU").
 *
 * This is the drop-in boundary for the hot path of paulmillr/noble-curves.  The reference
 * has no FFI seam - the seam is its public TypeScript surface - so each entry point below
 * names the reference function whose *result* it reproduces for a whole batch.  A thin
 * N-API addon or the Python host mirror (noble-curves_amd/) binds these.
 *
 * Wire format:
 *   - field elements: canonical residues (NOT Montgomery), little-endian bytes, fixed width:
 *       secp256k1 / ed25519 Fp: 32 bytes;  bls12-381 Fp: 48 bytes;  Fp2 = c0 || c1 (96 bytes)
 *   - affine points: x || y.  Infinity is (0,0) on Weierstrass curves
 *       (reference src/abstract/weierstrass.ts:716 fromAffine, :966 toAffine) and (0,1) on
 *       Edwards (src/abstract/edwards.ts:606); outputs also carry a separate is_inf byte.
 *   - scalars: 32 bytes little-endian, 0 <= k < 2^256 (range rules of the reference are
 *       enforced by the host shim before crossing: weierstrass.ts:904,920; curve.ts:398-404).
 *
 * Conventions: every function returns 0 on success or a negative ncg_status; no C++ type,
 * exception or torch type crosses this boundary.  Pointers suffixed _dev are device (HBM)
 * addresses valid on the context's GPU; all others are host pointers.  `stream` is a
 * hipStream_t passed as void* (NULL = the context's own stream); *_dev calls are
 * asynchronous on that stream, host-pointer calls return after the result is in host memory.
 * Device buffers must be 16-byte aligned (any hipMalloc / torch allocation is, and so is any whole-record
 * offset into one: records are 32 bytes or more); the kernels read and write them with 16-byte accesses.
 *
 * Threading and memory: a context is not thread-safe - use it from one host thread at a time (the
 * reference is single-threaded); several contexts per process are fine.  The context
 * owns its device workspaces (grown on demand, reused across calls, released by ncg_destroy): the
 * batch multiplies keep a Jacobian scratch plus a per-item window table (secp256k1 1.6 KB, ed25519
 * 1.1 KB, bls12-381 G1 1.5 KB / G2 3.0 KB per item), the MSM about 650 B per point at 2^20, the NTT
 * a twiddle table of N elements per transform size.  Because these buffers are shared, *_dev calls
 * on one context must be issued in stream order (same stream, or externally ordered); growing a
 * workspace synchronises the stream it was used on.
 */
#ifndef NCG_H
#define NCG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ncg_ctx ncg_ctx;

enum ncg_curve {
  NCG_SECP256K1 = 0,     /* src/secp256k1.ts:48-64   */
  NCG_ED25519 = 1,       /* src/ed25519.ts:49-65     */
  NCG_BLS12_381_G1 = 2,  /* src/bls12-381.ts:134-148 */
  NCG_BLS12_381_G2 = 3   /* src/bls12-381.ts:321-345 */
};

enum ncg_status {
  NCG_OK = 0,
  NCG_ERR_INVALID_ARG = -1,
  NCG_ERR_HIP = -2,
  NCG_ERR_NO_DEVICE = -3,
  NCG_ERR_UNSUPPORTED = -4,
  NCG_ERR_NOMEM = -5,
  NCG_ERR_RCCL = -6
};

/* ---- context ------------------------------------------------------------------------- */
/* One context per GPU.  Multi-GPU: see the "multi-GPU MSM" section below (per-process
 * communicators for one-process-per-GPU launches, ncg_multi for one process driving several GPUs). */
int ncg_init(int device_id, ncg_ctx** out_ctx);
void ncg_destroy(ncg_ctx* ctx);
const char* ncg_last_error(ncg_ctx* ctx); /* ctx may be NULL: last process-wide error */
int ncg_sync(ncg_ctx* ctx);               /* waits for the context's own stream */
const char* ncg_version(void);
/* bytes per affine point / per field element for a curve (0 if unknown) */
int ncg_point_bytes(int curve);
int ncg_field_bytes(int curve);

/* Pin a long-lived HOST buffer once (hipHostRegister): the host-pointer entry points then move it by DMA at PCIe speed
 * with no per-call page locking - what a binding does for buffers it reuses (the N-API addon's input / output Buffers).
 * Unpinned buffers stay correct: large ones are registered for the duration of each call. */
int ncg_host_register(void* host_ptr, size_t bytes);
int ncg_host_unregister(void* host_ptr);

/* ---- batch variable-base scalar multiplication ---------------------------------------
 * out[i] = scalars[i] * points[i].  Replaces, batch-wise, Point.multiplyUnsafe(k)
 * (src/abstract/weierstrass.ts:915-928; GLV path :660-671 on secp256k1) and the value of
 * Point.multiply(k) (:900-907); on ed25519 Point.multiplyUnsafe / multiply
 * (src/abstract/edwards.ts:555-577), exact integer multiples also for points with a torsion
 * component.  k = 0 or P = infinity gives infinity.  All four curves.  out_is_inf may be NULL
 * in the host-pointer variant. */
int ncg_mul_var_batch(ncg_ctx* ctx, int curve, size_t n, const void* points_affine,
                      const void* scalars, void* out_affine, uint8_t* out_is_inf);
int ncg_mul_var_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* points_affine_dev,
                          const void* scalars_dev, void* out_affine_dev,
                          uint8_t* out_is_inf_dev, void* stream);

/* ---- batch normalisation of projective points -----------------------------------------------
 * (X, Y, Z) -> affine (x, y) = (X/Z, Y/Z) for a batch with one field inversion per 8 points:
 * normalizeZ(c, points) / FpInvertBatch (src/abstract/curve.ts:311-326,
 * src/abstract/modular.ts:728-760, toAffine(invZ) src/abstract/weierstrass.ts:951-969; the same
 * map on Edwards extended points, src/abstract/edwards.ts:595-609).  Input wire: X || Y || Z,
 * canonical residues (3 field elements per point).  Z = 0 (Weierstrass identity) gives (0,0)
 * and out_is_inf = 1.  All four curves. */
int ncg_normalize_batch(ncg_ctx* ctx, int curve, size_t n, const void* points_proj, void* out_affine,
                        uint8_t* out_is_inf);
int ncg_normalize_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* points_proj_dev,
                            void* out_affine_dev, uint8_t* out_is_inf_dev, void* stream);

/* ---- batch point decoding (decompression + validity checks) ---------------------------------
 * out[i] = Point.fromBytes(encoded[i]) as an affine wire point, out_ok[i] = 0 where the reference
 * throws.  Encodings (fixed size per curve):
 *   NCG_SECP256K1     33 bytes SEC1 compressed (src/abstract/weierstrass.ts:566-605,
 *                     sqrt src/secp256k1.ts:73-95)
 *   NCG_BLS12_381_G1  48 bytes compressed with flag bits (src/bls12-381.ts:377-459); includes
 *                     the prime-order-subgroup check of assertValidity (:567-577); the
 *                     canonical infinity encoding gives out_is_inf = 1
 *   NCG_BLS12_381_G2  96 bytes compressed, x.c1 || x.c0 (src/bls12-381.ts:354-368, Fp2.sqrt
 *                     src/abstract/tower.ts:476-500) + the psi subgroup check (:599-601)
 *   NCG_ED25519       32 bytes (src/abstract/edwards.ts:405-436); flags bit 0 = zip215
 * out_is_inf may be NULL in the host-pointer variant. */
#define NCG_DECODE_ZIP215 1
int ncg_decode_points_batch(ncg_ctx* ctx, int curve, size_t n, const void* encoded, int flags,
                            void* out_affine, uint8_t* out_ok, uint8_t* out_is_inf);
int ncg_decode_points_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* encoded_dev, int flags,
                                void* out_affine_dev, uint8_t* out_ok_dev, uint8_t* out_is_inf_dev,
                                void* stream);

/* ---- pairwise point addition -------------------------------------------------------------------
 * out[i] = a[i] + b[i] (subtract != 0: a[i] - b[i]) on affine wire points: Point.add / subtract of the
 * reference for a batch of pairs, incl. P = Q, P = -Q and ZERO operands (src/abstract/weierstrass.ts:
 * 834-891, src/abstract/edwards.ts:526-545).  With two batch multiplies this is Point.mulAddUnsafe
 * (a*P + b*Q, weierstrass.ts:937-944 - the ECDSA-verification shape). */
int ncg_add_pairs_batch(ncg_ctx* ctx, int curve, size_t n, const void* a, const void* b, int subtract,
                        void* out_affine, uint8_t* out_is_inf);
int ncg_add_pairs_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* a_dev, const void* b_dev,
                            int subtract, void* out_affine_dev, uint8_t* out_is_inf_dev, void* stream);

/* ---- aggregation of encoded points -----------------------------------------------------------
 * out = sum_i Point.fromBytes(encoded[i]): the group operation of bls.aggregatePublicKeys /
 * aggregateSignatures on encoded inputs (src/abstract/bls.ts:857-873: normPub / fromBytes +
 * assertValidity, then the running sum).  Decoding (incl. the subgroup checks) and the sum - an MSM
 * with unit scalars - run on the device; only the encodings go up.  An entry the reference would
 * reject makes the call fail with NCG_ERR_INVALID_ARG and *out_bad_index = its position (else -1).
 * Encodings and `flags` as in ncg_decode_points_batch. */
int ncg_aggregate_encoded(ncg_ctx* ctx, int curve, size_t n, const void* encoded, int flags,
                          void* out_affine, uint8_t* out_is_inf, int64_t* out_bad_index);

/* ---- batch point encoding (Point.toBytes, compressed form) -----------------------------------
 * encoded[i] = affine wire point i in the encodings listed above (secp256k1 pointToBytes
 * src/abstract/weierstrass.ts:541-564; bls12-381 coder.encode src/bls12-381.ts:400-410; ed25519
 * src/abstract/edwards.ts:620-628).  out_ok[i] = 0 where the reference throws (secp256k1 ZERO). */
int ncg_encode_points_batch(ncg_ctx* ctx, int curve, size_t n, const void* affine, void* out_encoded,
                            uint8_t* out_ok);
int ncg_encode_points_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* affine_dev,
                                void* out_encoded_dev, uint8_t* out_ok_dev, void* stream);

/* ---- batch map-to-curve (hash-to-curve without the byte hashing) ------------------------------
 * out[i] = clearCofactor( sum_j mapToCurve(u[i][j]) ), j < count, for NCG_BLS12_381_G1 / _G2:
 * count = 2 is createHasher(...).hashToCurve after hash_to_field, count = 1 is encodeToCurve /
 * mapToCurve (src/abstract/hash-to-curve.ts:441-548; SWU :652-717, isogeny :381-410; suite
 * constants and clearCofactor src/bls12-381.ts:560-619, :668-862).  The byte-level
 * hash_to_field / expand_message_xmd (:189-228, :312-378) is done by the host shim.
 * u: n * count field elements, each one Fp (48 bytes LE) for G1 or Fp2 (c0 then c1, 96 bytes)
 * for G2, any value below 2^384 (reduced mod p like Fp.create).  ZERO gives out_is_inf = 1 and
 * (0, 0).  out_is_inf may be NULL in the host-pointer variant. */
int ncg_map_to_curve_batch(ncg_ctx* ctx, int curve, size_t n, int count, const void* u,
                           void* out_affine, uint8_t* out_is_inf);
int ncg_map_to_curve_batch_dev(ncg_ctx* ctx, int curve, size_t n, int count, const void* u_dev,
                               void* out_affine_dev, uint8_t* out_is_inf_dev, void* stream);

/* ---- number-theoretic transform over a scalar field ------------------------------------------
 * out = FFT(roots, Fr).direct(in, brpInput, brpOutput) / .inverse(...) of the reference
 * (src/abstract/fft.ts:518-577 over FFTCore :422-480; tables rootsOfUnity :230-312) for `batch`
 * polynomials of N = 2^log2n coefficients each, stored back to back.  Elements: canonical
 * residues, 32 bytes little-endian.  `omega` (HOST pointer, 32 bytes, canonical) is the primitive
 * N-th root roots.omega(log2n) = G^((r-1)/N); the table roots(log2n) is built on the device at
 * first use and cached per size.  The inverse transform walks the reversed table
 * (roots.inverse, :296-304) and scales by 1/N (:568-570).  in and out may alias. */
#define NCG_FIELD_BLS12_381_FR 0
#define NCG_NTT_INVERSE 1
#define NCG_NTT_BRP_INPUT 2
#define NCG_NTT_BRP_OUTPUT 4
#define NCG_NTT_MAX_LOG2N 28
int ncg_ntt(ncg_ctx* ctx, int field, int log2n, size_t batch, const void* omega, const void* in,
            void* out, int flags);
int ncg_ntt_dev(ncg_ctx* ctx, int field, int log2n, size_t batch, const void* omega,
                const void* in_dev, void* out_dev, int flags, void* stream);

/* ---- batch fixed-base scalar multiplication -----------------------------------------------
 * out[i] = scalars[i] * BASE.  Replaces, batch-wise, Point.BASE.multiply(k) / multiplyUnsafe(k)
 * through the cached window table (ScalarMultiplier.wnafCachedCT, src/abstract/curve.ts:588-606;
 * table :560-577) - e.g. getPublicKey.  k = 0 gives infinity.  The table (33 x 128 window
 * multiples of BASE) is built on the device at first use and cached in the context. */
int ncg_mul_base_batch(ncg_ctx* ctx, int curve, size_t n, const void* scalars, void* out_affine,
                       uint8_t* out_is_inf);
int ncg_mul_base_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* scalars_dev,
                           void* out_affine_dev, uint8_t* out_is_inf_dev, void* stream);

/* ---- multi-scalar multiplication ---------------------------------------------------------
 * out = sum_i scalars[i] * points[i].  Replaces pippenger(c, points, scalars)
 * (src/abstract/curve.ts:863-905) on all four curves; n = 0 gives infinity (:878); scalar 0 and
 * infinity points are allowed.  The result (one affine point, canonical residues) is written to HOST memory
 * in both variants: the last step (Horner over <= 272 window/level sums + one inversion) runs
 * on the host, so the call returns after synchronising the stream. */
int ncg_msm(ncg_ctx* ctx, int curve, size_t n, const void* points_affine, const void* scalars,
            void* out_affine, uint8_t* out_is_inf);
/* The window plan the MSM entry points use for n points (measurement / diagnostics): out4 = {window bits c,
 * windows, buckets per window 2^(c-1) (signed digits), grouped window sums per window handed to the host finish}. */
int ncg_msm_plan_info(int curve, size_t n, int* out4);
int ncg_msm_dev(ncg_ctx* ctx, int curve, size_t n, const void* points_affine_dev,
                const void* scalars_dev, void* out_affine, uint8_t* out_is_inf, void* stream);
/* Diagnostics of the MSM pipeline on one context.  ncg_msm_set_tuning overrides two internals that decide HOW buckets are
 * cut, never the result: `seg` = sorted entries per accumulate lane (lane boundaries cut buckets into pieces), `run_serial`
 * = how many following pieces the owner of a cut bucket adds itself before the run goes to the long-run work list
 * (k_msm_fixup_long); seg <= 0 / run_serial < 0 restore the measured defaults.  ncg_msm_last_plan reports what the last MSM
 * launch on the context ran with: out8 = {window bits c, windows run, buckets per window, first window, windows of the whole
 * plan, seg, run_serial, runs that went to the work list}; it synchronises the device.  (tests/test_gpu_msm.py uses the pair
 * to cut buckets in every possible way and to prove that the requested cut was the one executed.) */
int ncg_msm_set_tuning(ncg_ctx* ctx, int seg, int run_serial);
int ncg_msm_last_plan(ncg_ctx* ctx, int* out8);

/* ---- resident point sets --------------------------------------------------------------------
 * Upload a point set once, multiply many times with only the scalars crossing - the usage pattern of
 * the reference's interleavedMSMUnsafe closure (src/abstract/curve.ts:907-959: precompute for a fixed
 * point set, call with scalars), and the answer to SURVEY 8a gotcha 8 (marshalling dominates an
 * end-to-end call).  ncg_points_from_encoded takes the points in their compressed wire encodings
 * (sizes as for ncg_decode_points_batch) and decodes / validates them on the device; an entry the
 * reference's fromBytes would reject fails the call and is named in *out_bad_index.  A handle
 * belongs to the context it was made on; free it before the context. */
typedef struct ncg_points ncg_points;
int ncg_points_upload(ncg_ctx* ctx, int curve, size_t n, const void* points_affine, ncg_points** out);
int ncg_points_from_encoded(ncg_ctx* ctx, int curve, size_t n, const void* encoded, int flags,
                            ncg_points** out, int64_t* out_bad_index);
void ncg_points_free(ncg_points* pts);
size_t ncg_points_count(const ncg_points* pts);
int ncg_points_curve(const ncg_points* pts);
const void* ncg_points_dev(const ncg_points* pts); /* device address of the affine wire points */
/* bls12-381 G1 / G2 sets whose points all lie in the prime-order subgroup get a faster ncg_msm_resident: the
 * scalars are split along the curve endomorphism (G1: k = k1 + k2 z^2 with phi, the map of the reference's own
 * subgroup test bls12-381.ts:567-577; G2: k = sum d_e z^e with psi, :599-601), which halves / quarters the
 * window count for the same group element.  pippenger itself (curve.ts:863-905) takes arbitrary curve points,
 * so this is never assumed: sets from ncg_points_from_encoded qualify automatically (the decoder runs the
 * reference's subgroup test on every point); for uploaded affine points ncg_points_verify_subgroup runs the
 * test once - every point P must satisfy [z^2]P = -phi(P) resp. [z]P = -psi(P) - and enables the path only if
 * all pass.  A failing set is not an error: *out_bad_index names the first point outside the subgroup and the
 * set keeps using the generic path.  ncg_points_in_subgroup: 1 if the fast path is active.  A verified set also
 * makes ncg_mul_var_batch_resident* use the endomorphism ladders (G1: two 128-bit half-scalars, half the doublings;
 * G2: four 64-bit streams along psi, a quarter of the doublings) - same group elements as the generic ladder. */
int ncg_points_verify_subgroup(ncg_ctx* ctx, ncg_points* pts, int64_t* out_bad_index);
/* The precomputation of interleavedMSMUnsafe (src/abstract/curve.ts:907-959: per-point tables built ONCE for a
 * fixed point set) in device form: window-shifted copies 2^(16 w) P of every point (or of every endomorphism image
 * of a verified set - call ncg_points_verify_subgroup first to get those), 16 / 8 / 4 copies of the set in device
 * memory.  ncg_msm_resident* then adds every window into ONE bucket set: the bucket fold runs once instead of once
 * per window and the serial combine across windows (curve.ts:901-902) disappears.  Same group element, bit for bit.
 * Weierstrass curves, sets of >= 4096 points; otherwise (and when memory is short) the call succeeds and changes
 * nothing.  ncg_points_precomputed: 1 if the shared-bucket path is active. */
int ncg_points_precompute(ncg_ctx* ctx, ncg_points* pts);
int ncg_points_precomputed(const ncg_points* pts);
int ncg_points_in_subgroup(const ncg_points* pts);
/* pippenger(c, <resident points>, scalars) and the batch multiplyUnsafe on them; scalars: host */
int ncg_msm_resident(ncg_ctx* ctx, const ncg_points* pts, const void* scalars, void* out_affine,
                     uint8_t* out_is_inf);
int ncg_msm_resident_dev(ncg_ctx* ctx, const ncg_points* pts, const void* scalars_dev, void* out_affine,
                         uint8_t* out_is_inf, void* stream); /* scalars: device, 32 B LE each */
int ncg_mul_var_batch_resident(ncg_ctx* ctx, const ncg_points* pts, const void* scalars,
                               void* out_affine, uint8_t* out_is_inf);
/* the same with scalars (32 B LE each), results and infinity flags (all required) in device memory */
int ncg_mul_var_batch_resident_dev(ncg_ctx* ctx, const ncg_points* pts, const void* scalars_dev,
                                   void* out_affine_dev, uint8_t* out_is_inf_dev, void* stream);

/* ---- multi-GPU MSM ---------------------------------------------------------------------
 * pippenger is a sum over points (src/abstract/curve.ts:863-905; its last step is the chain
 * `sum = sum.add(resI)` :895-902), so the points are sharded: each GPU runs the single-GPU pipeline on
 * its slice up to the grouped window sums (~18 KB for G1), ONE ncclAllGather over xGMI exchanges those,
 * a small kernel adds the G arrays (pairwise tree, log2 G additions deep) and the usual finish follows.  No
 * bucket-sized data moves and nothing but the final point reaches the host.  All four curves
 * (BASELINE configs[3] G1 and configs[4] G2).  SURVEY 8(b): "one RCCL communicator for the device
 * set, all hidden behind the call"; RCCL (librccl.so) is loaded on first use.
 *
 * (1) One process per GPU (torch.distributed, MPI, bench.py): rank 0 calls ncg_comm_unique_id and
 *     ships the 128 bytes to the other ranks by any means; every rank calls ncg_comm_init on its own
 *     context (collective).  ncg_msm_sharded_dev is then a collective: every rank passes ITS points
 *     (n_local of them, device memory) and n_max = the largest n_local of any rank (it fixes the
 *     window plan, which must agree on all ranks; pass 0 only if all ranks hold n_local points); every
 *     rank receives the same sum in host memory.  Without a communicator it is ncg_msm_dev.
 *     Every rank posts a slot of the same fixed size (ncg_msm_shard_slot_bytes) whatever plan it derived;
 *     ranks whose plans disagree get NCG_ERR_INVALID_ARG after the gather, never a mismatched collective. */
#define NCG_COMM_ID_BYTES 128
int ncg_comm_unique_id(uint8_t* out_id128);
int ncg_comm_init(ncg_ctx* ctx, int nranks, int rank, const uint8_t* id128);
int ncg_comm_destroy(ncg_ctx* ctx); /* also done by ncg_destroy */
int ncg_comm_size(ncg_ctx* ctx);
int ncg_comm_rank(ncg_ctx* ctx);
/* the communicator's OWN answer (ncclCommCount / ncclCommUserRank); *out_ranks = 0 when the context has none */
int ncg_comm_count(ncg_ctx* ctx, int* out_ranks, int* out_rank);
int ncg_msm_sharded_dev(ncg_ctx* ctx, int curve, size_t n_local, size_t n_max,
                        const void* points_affine_dev, const void* scalars_dev, void* out_affine,
                        uint8_t* out_is_inf, void* stream);
/* The same sharded MSM with a HOST-STAGED exchange, for transports other than RCCL (gloo, MPI, sockets; also how
 * two ranks sharing one GPU are tested): ncg_msm_shard_local_dev runs this rank's per-shard phase and writes its
 * slot - a 32-byte header (window plan, window range, mode, scalar-range verdict) + the grouped window sums,
 * ncg_msm_shard_slot_bytes(curve) bytes, zero padded - to host memory; a scalar outside the group order does NOT fail
 * this call: the verdict travels in the header and fails ncg_msm_shard_combine on EVERY rank (so no rank leaves the
 * exchange early); the caller gathers the slots of all ranks (rank order) and any rank calls
 * ncg_msm_shard_combine on the concatenation: upload, header check, adding kernel, finish - the code
 * ncg_msm_sharded_dev runs after its all-gather.  n_max as above (the same value on every rank and in combine). */
size_t ncg_msm_shard_slot_bytes(int curve);
int ncg_msm_shard_local_dev(ncg_ctx* ctx, int curve, size_t n_local, size_t n_max, const void* points_affine_dev,
                            const void* scalars_dev, void* slot_out, void* stream);
int ncg_msm_shard_combine(ncg_ctx* ctx, int curve, size_t n_max, int nparts, const void* slots, void* out_affine,
                          uint8_t* out_is_inf, void* stream);
/* WINDOW-sharded mode (strong scaling of ONE MSM; round 4).  pippenger's windows are independent until its final
 * double-and-add chain (src/abstract/curve.ts:886-902), so when every rank holds ALL n points and scalars - a replicated
 * array, or a resident set uploaded on every GPU (the fixed bases of a prover) - rank r of G runs only a contiguous range of
 * the windows (2 of 16 at G = 8 for 2^20 bls12-381 G1 points): digits, sort, accumulate and the throughput part of the
 * bucket fold all divide by G, where point sharding leaves every rank the full fold and tail.  The slots hold the ranks'
 * grouped window sums; they are CONCATENATED (no group additions) and every rank runs the Horner chain.  On a
 * precomputed resident set (ncg_points_precompute) the rank's windows add into ONE bucket set and the slots are added.
 * Pass points_affine_dev (n points, device) or `resident` (then curve / n / points_affine_dev are ignored); scalars_dev:
 * all n scalars.  Collective over the context's communicator like ncg_msm_sharded_dev; without one it is a single-GPU MSM.
 * Host-staged twin: ncg_msm_shard_windows_local_dev (part r of nparts) -> any all-gather -> ncg_msm_shard_combine (which
 * reads the mode from the slot headers). */
int ncg_msm_sharded_windows_dev(ncg_ctx* ctx, int curve, size_t n, const void* points_affine_dev, const ncg_points* resident,
                                const void* scalars_dev, void* out_affine, uint8_t* out_is_inf, void* stream);
int ncg_msm_shard_windows_local_dev(ncg_ctx* ctx, int curve, size_t n, int part, int nparts, const void* points_affine_dev,
                                    const ncg_points* resident, const void* scalars_dev, void* slot_out, void* stream);
/* the window-sharded pipeline on ONE GPU: the parts run in turn, then the same concatenation / finish */
int ncg_msm_split_windows_dev(ncg_ctx* ctx, int curve, size_t n, int parts, const void* points_affine_dev,
                              const ncg_points* resident, const void* scalars_dev, void* out_affine, uint8_t* out_is_inf,
                              void* stream);
/* Several MSMs in flight (round 4).  A context has ncg_msm_async_lanes() lanes; each owns a stream, a workspace and a
 * pinned landing area.  ncg_msm_async_submit enqueues one MSM on a lane and returns at once (`stream`, if given, is the
 * stream the inputs were produced on; the lane waits for it); ncg_msm_async_collect waits for that MSM, runs the host
 * finish and frees the lane.  Used in turn, the lanes overlap the dependent tail of one MSM (narrow fold levels, per-window
 * tail, D2H, host Horner: latency, not throughput) with the sort / accumulate kernels of the next - the steady-state time
 * per MSM is the throughput part alone.  Points: device array or `resident` (as above).  flags: NCG_MSM_ASYNC_WINDOWS =
 * window-sharded over the context's communicator (collective: all ranks submit / collect the same lanes in the same order).
 * Errors of the MSM itself (scalar outside the group order, plans that disagree) are reported by collect. */
#define NCG_MSM_ASYNC_WINDOWS 1
/* ONE part of a window-sharded MSM on a lane (part p of P, no communicator): its slot comes back through
 * ncg_msm_async_collect_slot and goes, with the other parts' slots, to ncg_msm_shard_combine.  This is the host-staged
 * exchange with several parts in flight - and how one GPU measures a rank's share of a G-GPU MSM (bench.py). */
#define NCG_MSM_ASYNC_PART_FLAG 2
#define NCG_MSM_ASYNC_PART(p, P) (NCG_MSM_ASYNC_PART_FLAG | ((p) << 8) | ((P) << 20))
int ncg_msm_async_collect_slot(ncg_ctx* ctx, int lane, void* slot_out);
int ncg_msm_async_lanes(void);
int ncg_msm_async_submit(ncg_ctx* ctx, int lane, int curve, size_t n, const void* points_affine_dev, const ncg_points* resident,
                         const void* scalars_dev, int flags, void* stream);
int ncg_msm_async_collect(ncg_ctx* ctx, int lane, void* out_affine, uint8_t* out_is_inf);
/* The sharded pipeline on ONE GPU (self-check, shard-plan A/B): the points are cut into `parts` slices,
 * each runs the per-shard phase in turn, the slices' window sums go through the multi-GPU combine kernel
 * and finish - everything of ncg_msm_sharded_dev except the all-gather. */
int ncg_msm_split_dev(ncg_ctx* ctx, int curve, size_t n, int parts, const void* points_affine_dev,
                      const void* scalars_dev, void* out_affine, uint8_t* out_is_inf, void* stream);
/* (2) One process, several GPUs (the shape of the N-API addon: Node runs one thread).  ncg_multi_init
 *     opens a context per device and one communicator over the set (ncclCommInitAll); ncg_msm_multi
 *     takes HOST arrays like ncg_msm, uploads slice g to device g, and returns the sum. */
typedef struct ncg_multi ncg_multi;
int ncg_multi_init(const int* device_ids, int n_dev, ncg_multi** out);
void ncg_multi_destroy(ncg_multi* m);
int ncg_multi_devices(ncg_multi* m);
ncg_ctx* ncg_multi_ctx(ncg_multi* m, int i); /* context of device i (for the single-GPU entry points) */
const char* ncg_multi_last_error(ncg_multi* m);
int ncg_msm_multi(ncg_multi* m, int curve, size_t n, const void* points_affine, const void* scalars,
                  void* out_affine, uint8_t* out_is_inf);

/* ---- ed25519 batch signature verification -------------------------------------------------
 * out_ok[i] = eddsa.verify(sig[i], msg[i], pk[i], {zip215}) of the reference
 * (src/abstract/edwards.ts:942-989) given the already-hashed challenge
 * k32[i] = SHA-512(R || A || M) mod L as 32 bytes LE (edwards.ts:984, :900-906; the hash is in
 * @noble/hashes and is computed by the host shim).  sig64 = R (32 B) || s (32 B) as on the wire;
 * pk32 = compressed public key.  zip215 != 0 selects the reference's default (ZIP-215) decoding
 * rules, 0 the strict RFC 8032 rules (edwards.ts:405-436, :980).  Decoding failures and
 * s >= L give 0, exactly where the reference returns false (:967-975). */
int ncg_ed25519_verify_batch(ncg_ctx* ctx, size_t n, const void* sig64, const void* pk32,
                             const void* k32, int zip215, uint8_t* out_ok);
int ncg_ed25519_verify_batch_dev(ncg_ctx* ctx, size_t n, const void* sig64_dev,
                                 const void* pk32_dev, const void* k32_dev, int zip215,
                                 uint8_t* out_ok_dev, void* stream);

/* The same from the MESSAGES: the challenge k = SHA-512(R || A || M) mod L (edwards.ts:984, :900-906,
 * modN_LE :866-868; SHA-512 is @noble/hashes in the reference) is computed on the device, one lane per
 * signature, so nothing but (signature, key, message) crosses - SURVEY 8(b) `k32* | (msgs*, msg_off*)`.
 * msgs: the messages back to back; msg_off: n + 1 byte offsets, message i = msgs[msg_off[i] .. msg_off[i+1]).
 * ncg_ed25519_challenge_batch_dev exposes the hash step alone (out: n x 32 bytes LE). */
int ncg_ed25519_verify_batch_msgs(ncg_ctx* ctx, size_t n, const void* sig64, const void* pk32,
                                  const void* msgs, const uint64_t* msg_off, int zip215, uint8_t* out_ok);
int ncg_ed25519_verify_batch_msgs_dev(ncg_ctx* ctx, size_t n, const void* sig64_dev, const void* pk32_dev,
                                      const void* msgs_dev, const uint64_t* msg_off_dev, int zip215,
                                      uint8_t* out_ok_dev, void* stream);
int ncg_ed25519_challenge_batch_dev(ncg_ctx* ctx, size_t n, const void* sig64_dev, const void* pk32_dev,
                                    const void* msgs_dev, const uint64_t* msg_off_dev, void* out_k32_dev,
                                    void* stream);

/* ---- measurement helpers (not on the product path) ------------------------------------ */
/* ---- secp256k1 ECDSA batch verification --------------------------------------------------------
 * out_ok[i] = ecdsa.verify(sig[i], msgHash[i], publicKey[i], { prehash: false, format: 'compact', lowS })
 * (src/abstract/weierstrass.ts:1571-1620; the caller above Point.mulAddUnsafe, :1609).  sig: r || s, 32
 * big-endian bytes each (Signature.fromBytes 'compact': both in [1, n), else false); msgHash: 32 bytes, taken as
 * h = bits2int_modN; publicKey: 33-byte SEC1 compressed, or 65-byte uncompressed rows with NCG_ECDSA_PUB_UNCOMPRESSED
 * (Point.fromBytes; a key the reference rejects or the identity gives false).  flags: NCG_ECDSA_LOW_S = the reference's default lowS rule (s <= n/2).  Everything -
 * key decompression, s^-1 mod n (one inversion per 16 signatures), u1 G + u2 P, R.x mod n == r - runs on the
 * device.  NCG_SECP256K1 only. */
#define NCG_ECDSA_LOW_S 1
#define NCG_ECDSA_PUB_UNCOMPRESSED 2 /* public keys are 65-byte uncompressed SEC1 rows (04 || x || y): prefix, range and
                                        curve-equation checks on the device (weierstrass.ts:589-597), no square root */
int ncg_ecdsa_verify_batch(ncg_ctx* ctx, int curve, size_t n, const void* sig64, const void* hash32,
                           const void* pub33, int flags, uint8_t* out_ok);
int ncg_ecdsa_verify_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* sig64_dev,
                               const void* hash32_dev, const void* pub33_dev, int flags,
                               uint8_t* out_ok_dev, void* stream);

/* Public-key recovery for a batch: out[i] = Signature.fromBytes(sig65[i], 'recovered').recoverPublicKey(msgHash[i])
 * (src/abstract/weierstrass.ts:1391-1407) as an affine wire point.  sig65: recovery id (0..3) || r || s, 32 big-endian
 * bytes each; out_ok[i] = 0 where the reference throws (recid > 3, r or s outside [1, n), r + n >= p for recid 2 / 3,
 * no point with that x, Q = O) and the point is then zeroed.  NCG_SECP256K1 only. */
int ncg_ecdsa_recover_batch(ncg_ctx* ctx, int curve, size_t n, const void* sig65, const void* hash32,
                            void* out_affine, uint8_t* out_ok);
int ncg_ecdsa_recover_batch_dev(ncg_ctx* ctx, int curve, size_t n, const void* sig65_dev,
                                const void* hash32_dev, void* out_affine_dev, uint8_t* out_ok_dev, void* stream);

/* ---- secp256k1 BIP-340 Schnorr batch verification ----------------------------------------------
 * out_ok[i] = schnorr.verify(sig[i], msg[i], publicKey[i]) (src/secp256k1.ts:228-258) given the challenge
 * e[i] = int(taggedHash('BIP0340/challenge', r || pk || msg)) mod n as 32 big-endian bytes (:176-178; the host
 * shim hashes, like the reference's host sha256).  sig: r || s big-endian (r in [1, p), s in [1, n), else false);
 * publicKey: 32-byte x-only, P = lift_x (:158-170: the even root; no root / x >= p gives false).  On the device:
 * the range checks, lift_x, R = s G + (n - e) P, and R != O, even y(R), x(R) == r. */
int ncg_schnorr_verify_batch(ncg_ctx* ctx, size_t n, const void* sig64, const void* e32, const void* pkx32,
                             uint8_t* out_ok);
int ncg_schnorr_verify_batch_dev(ncg_ctx* ctx, size_t n, const void* sig64_dev, const void* e32_dev,
                                 const void* pkx32_dev, uint8_t* out_ok_dev, void* stream);

/* ---- the two verifications from MESSAGES (hash on the device, csrc/sha256.hpp) -------------------------------
 * ncg_ecdsa_verify_batch_msgs: ecdsa.verify(sig, msg, publicKey, { prehash: true, ... }) - SHA-256(msg) is the message
 * representative (weierstrass.ts:1465-1470); ncg_schnorr_verify_batch_msgs: schnorr.verify(sig, msg, pk) with the tagged
 * challenge hash (src/secp256k1.ts:129-137, :176-178).  msgs = all messages back to back, msg_off = n + 1 byte
 * offsets (message i = msgs[msg_off[i] .. msg_off[i+1])); everything else as in the entry points above. */
int ncg_ecdsa_verify_batch_msgs(ncg_ctx* ctx, int curve, size_t n, const void* sig64, const void* msgs,
                                const uint64_t* msg_off, const void* pub, int flags, uint8_t* out_ok);
int ncg_ecdsa_verify_batch_msgs_dev(ncg_ctx* ctx, int curve, size_t n, const void* sig64_dev, const void* msgs_dev,
                                    const uint64_t* msg_off_dev, const void* pub_dev, int flags,
                                    uint8_t* out_ok_dev, void* stream);
int ncg_schnorr_verify_batch_msgs(ncg_ctx* ctx, size_t n, const void* sig64, const void* msgs,
                                  const uint64_t* msg_off, const void* pkx32, uint8_t* out_ok);
int ncg_schnorr_verify_batch_msgs_dev(ncg_ctx* ctx, size_t n, const void* sig64_dev, const void* msgs_dev,
                                      const uint64_t* msg_off_dev, const void* pkx32_dev, uint8_t* out_ok_dev,
                                      void* stream);

/* Runs instruction-rate / field-multiply micro-benchmark `kind` (see csrc/ubench.hip) and
 * returns the kernel time in milliseconds. */
int ncg_ubench(ncg_ctx* ctx, int kind, int blocks, int threads, int iters, float* out_ms);
/* Field-level self-check: out[i] = op(a[i], b[i]) computed by the device field code, one lane per element
 * (host buffers).  field 0 / 1 = secp256k1 / ed25519 base field in the radix-2^29 lazy form (fe9.hpp): a, b
 * are 9 RAW 32-bit limbs per element, `variant` = 10 A + B names the operand bound types; out = 8 canonical
 * LE words.  field 2 = bls12-381 Fp: 12-word canonical operands and results.  op: 0 mul, 1 sqr, 2 add, 3 sub,
 * 4 neg, 5 inv, 6 normalise, 7 is-zero-mod-p, 9 largest output limb of the product.
 * field 3 = bls12-381 Fp on RAW radix-2^29 limbs (fe29.hpp, Montgomery R = 2^406): a = [a, c], b = [b, d], 14 limbs
 * each, values up to the top of the lazy bounds (a, c < 4096 p; b, d < 4096 p for op 0, 2048 p for op 6);
 * field 4 = the lane-paired Fp2 form: every element c0 then c1 (28 limbs), a, c < 4096 p (2048 p for op 1), b, d <
 * 2048 p (op 0) / 1024 p (op 6).  ops 0 a*b, 1 a^2, 6 a*b - c*d (the fused reduction); out = canonical wire words
 * of the result taken out of Montgomery form (12, or 12 + 12 for Fp2). */
int ncg_field_check(ncg_ctx* ctx, int field, int op, int variant, size_t n, const void* a, const void* b, void* out);

#ifdef __cplusplus
}
#endif
#endif /* NCG_H */
