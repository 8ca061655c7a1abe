// N-API addon: the binding a noble-curves maintainer would add to reach libncg.so from the
// reference's TypeScript/JavaScript side.  Synchronous, like every call on
// the reference's path (there is no async anywhere in src/abstract/curve.ts).
//
//   init(deviceId = 0)                               -> undefined (throws Error('noble-gpu: ...'))
//   msm(curveId, points: Uint8Array, scalars: Uint8Array)            -> Uint8Array PB + 1 (flag)
//   mulVarBatch(curveId, points, scalars[, out])     -> Uint8Array n * (PB + 1)  (points then flags; `out`: reuse the caller's array)
//   mulBaseBatch(curveId, scalars)                   -> Uint8Array n * (PB + 1)
//   ed25519VerifyBatch(sigs, pks, ks, zip215: bool)  -> Uint8Array n (0 / 1)
//   decodePoints(curveId, encoded, zip215: bool)     -> Uint8Array n * (PB + 2)  (points, ok flags, inf flags)
//   encodePoints(curveId, points)                    -> Uint8Array n * (EB + 1)  (encodings, ok flags)
//   aggregateEncoded(curveId, encoded, zip215)       -> Uint8Array PB + 1 (flag); throws naming a bad index
//   ntt(log2n, omega: Uint8Array 32, data, flags)    -> Uint8Array (same length)
//   mapToCurve(curveId, count, u)                    -> Uint8Array n * (PB + 1)
//   pointBytes(curveId) / version()
// Buffers use the wire format of include/ncg.h.  Build: make -C addon  (g++ + /usr/include/node).
#include <node_api.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../include/ncg.h"

static ncg_ctx* g_ctx = nullptr;

#define NAPI_OK(call)                                             \
  do {                                                            \
    if ((call) != napi_ok) {                                      \
      napi_throw_error(env, nullptr, "noble-gpu: N-API failure"); \
      return nullptr;                                             \
    }                                                             \
  } while (0)

static napi_value throw_native(napi_env env) {
  const char* m = ncg_last_error(g_ctx);
  napi_throw_error(env, nullptr, (m && *m) ? m : "noble-gpu: native call failed");
  return nullptr;
}

static bool get_u8(napi_env env, napi_value v, uint8_t** data, size_t* len) {
  bool is_ta = false;
  if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta) return false;
  napi_typedarray_type t;
  napi_value ab;
  size_t off;
  void* p;
  if (napi_get_typedarray_info(env, v, &t, len, &p, &ab, &off) != napi_ok || t != napi_uint8_array) return false;
  *data = (uint8_t*)p;
  return true;
}

static napi_value make_u8(napi_env env, size_t n, uint8_t** out) {
  napi_value ab, arr;
  void* p;
  if (napi_create_arraybuffer(env, n, &p, &ab) != napi_ok) return nullptr;
  if (napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &arr) != napi_ok) return nullptr;
  *out = (uint8_t*)p;
  return arr;
}

static napi_value Init(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  int32_t dev = 0;
  if (argc >= 1) napi_get_value_int32(env, argv[0], &dev);
  if (!g_ctx && ncg_init(dev, &g_ctx) != 0) {
    napi_throw_error(env, nullptr, ncg_last_error(nullptr));
    return nullptr;
  }
  return nullptr;
}

// initMulti([deviceIds]): one process, several GPUs (include/ncg.h "multi-GPU MSM" (2)).  The context of the
// first device serves the single-GPU entry points; msm() shards over the whole set.
static ncg_multi* g_multi = nullptr;
static napi_value InitMulti(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  bool is_arr = false;
  uint32_t n = 0;
  if (argc < 1 || napi_is_array(env, argv[0], &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, argv[0], &n) != napi_ok ||
      n == 0 || n > 64) {
    napi_throw_type_error(env, nullptr, "noble-gpu: initMulti([deviceId, ...])");
    return nullptr;
  }
  if (g_ctx || g_multi) {
    napi_throw_error(env, nullptr, "noble-gpu: already initialised");
    return nullptr;
  }
  int ids[64];
  for (uint32_t i = 0; i < n; i++) {
    napi_value e;
    int32_t v = 0;
    if (napi_get_element(env, argv[0], i, &e) != napi_ok || napi_get_value_int32(env, e, &v) != napi_ok) {
      napi_throw_type_error(env, nullptr, "noble-gpu: initMulti: device ids must be integers");
      return nullptr;
    }
    ids[i] = v;
  }
  if (ncg_multi_init(ids, (int)n, &g_multi) != 0) {
    napi_throw_error(env, nullptr, ncg_last_error(nullptr));
    return nullptr;
  }
  g_ctx = ncg_multi_ctx(g_multi, 0);
  napi_value r;
  napi_create_uint32(env, (uint32_t)ncg_multi_devices(g_multi), &r);
  return r;
}

static bool need_ctx(napi_env env) {
  if (g_ctx) return true;
  napi_throw_error(env, nullptr, "noble-gpu: call init() first");
  return false;
}

static napi_value Msm(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  int32_t curve;
  uint8_t *pts, *sc, *out;
  size_t pl, sl;
  if (argc < 3 || napi_get_value_int32(env, argv[0], &curve) != napi_ok || !get_u8(env, argv[1], &pts, &pl) ||
      !get_u8(env, argv[2], &sc, &sl)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: msm(curveId, Uint8Array, Uint8Array)");
    return nullptr;
  }
  int pb = ncg_point_bytes(curve);
  if (pb == 0 || sl % 32 || pl != (sl / 32) * (size_t)pb) {
    napi_throw_error(env, nullptr, "arrays of points and scalars must have equal length");
    return nullptr;
  }
  napi_value res = make_u8(env, pb + 1, &out);
  if (!res) return nullptr;
  if (g_multi) {  // shard over the device set
    if (ncg_msm_multi(g_multi, curve, sl / 32, pts, sc, out, out + pb) != 0) {
      napi_throw_error(env, nullptr, ncg_multi_last_error(g_multi));
      return nullptr;
    }
    return res;
  }
  if (ncg_msm(g_ctx, curve, sl / 32, pts, sc, out, out + pb) != 0) return throw_native(env);
  return res;
}

// mulVarBatch(curveId, points, scalars[, out]): with `out` (a Uint8Array of n * (PB + 1) bytes the caller keeps - and may pin once
// with hostRegister) the results land there and `out` is returned; without it a fresh array is made per call, which at 2^20 items is
// 68 MB of first-touch page faults and a per-call page lock before the first result byte can be written.
static napi_value MulVarBatch(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  int32_t curve;
  uint8_t *pts, *sc, *out;
  size_t pl, sl;
  if (argc < 3 || napi_get_value_int32(env, argv[0], &curve) != napi_ok || !get_u8(env, argv[1], &pts, &pl) ||
      !get_u8(env, argv[2], &sc, &sl)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: mulVarBatch(curveId, Uint8Array, Uint8Array[, Uint8Array out])");
    return nullptr;
  }
  int pb = ncg_point_bytes(curve);
  size_t n = sl / 32;
  if (pb == 0 || sl % 32 || pl != n * (size_t)pb) {
    napi_throw_error(env, nullptr, "arrays of points and scalars must have equal length");
    return nullptr;
  }
  napi_value res;
  napi_valuetype vt = napi_undefined;
  if (argc >= 4) napi_typeof(env, argv[3], &vt);
  if (argc >= 4 && vt != napi_undefined && vt != napi_null) {
    size_t ol;
    if (!get_u8(env, argv[3], &out, &ol) || ol != n * (size_t)(pb + 1)) {
      napi_throw_error(env, nullptr, "noble-gpu: mulVarBatch: out must be a Uint8Array of n * (point bytes + 1) bytes");
      return nullptr;
    }
    res = argv[3];
  } else {
    res = make_u8(env, n * (pb + 1), &out);
    if (!res) return nullptr;
  }
  if (n && ncg_mul_var_batch(g_ctx, curve, n, pts, sc, out, out + n * pb) != 0) return throw_native(env);
  return res;
}

static napi_value MulBaseBatch(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  int32_t curve;
  uint8_t *sc, *out;
  size_t sl;
  if (argc < 2 || napi_get_value_int32(env, argv[0], &curve) != napi_ok || !get_u8(env, argv[1], &sc, &sl) || sl % 32) {
    napi_throw_type_error(env, nullptr, "noble-gpu: mulBaseBatch(curveId, Uint8Array)");
    return nullptr;
  }
  int pb = ncg_point_bytes(curve);
  size_t n = sl / 32;
  napi_value res = make_u8(env, n * (pb + 1), &out);
  if (!res) return nullptr;
  if (n && ncg_mul_base_batch(g_ctx, curve, n, sc, out, out + n * pb) != 0) return throw_native(env);
  return res;
}

static napi_value Ed25519VerifyBatch(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  uint8_t *sig, *pk, *k, *out;
  size_t sgl, pkl, kl;
  bool zip215 = true;
  if (argc < 3 || !get_u8(env, argv[0], &sig, &sgl) || !get_u8(env, argv[1], &pk, &pkl) || !get_u8(env, argv[2], &k, &kl)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: ed25519VerifyBatch(sigs, pks, ks, zip215)");
    return nullptr;
  }
  if (argc >= 4) napi_get_value_bool(env, argv[3], &zip215);
  size_t n = sgl / 64;
  if (sgl % 64 || pkl != n * 32 || kl != n * 32) {
    napi_throw_error(env, nullptr, "arrays of signatures, public keys and challenges must have equal length");
    return nullptr;
  }
  napi_value res = make_u8(env, n, &out);
  if (!res) return nullptr;
  if (n && ncg_ed25519_verify_batch(g_ctx, n, sig, pk, k, zip215 ? 1 : 0, out) != 0) return throw_native(env);
  return res;
}

// eddsa.verify from (sig, msg, pk): the challenge hash runs on the device too.  msgs = all messages back to
// back, offs = Float64Array / BigUint64Array-free: a Uint8Array holding n + 1 little-endian uint64 offsets.
static napi_value Ed25519VerifyMsgs(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value argv[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  uint8_t *sig, *pk, *msgs, *offs, *out;
  size_t sgl, pkl, ml, ol;
  bool zip215 = true;
  if (argc < 4 || !get_u8(env, argv[0], &sig, &sgl) || !get_u8(env, argv[1], &pk, &pkl) || !get_u8(env, argv[2], &msgs, &ml) ||
      !get_u8(env, argv[3], &offs, &ol)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: ed25519VerifyMsgs(sigs, pks, msgs, offsets, zip215)");
    return nullptr;
  }
  if (argc >= 5) napi_get_value_bool(env, argv[4], &zip215);
  const size_t n = sgl / 64;
  if (sgl % 64 || pkl != n * 32 || ol != (n + 1) * 8) {
    napi_throw_error(env, nullptr, "arrays of signatures, public keys and message offsets must have matching lengths");
    return nullptr;
  }
  std::vector<uint64_t> off(n + 1);
  memcpy(off.data(), offs, ol);
  if (off[n] > ml) {
    napi_throw_error(env, nullptr, "noble-gpu: message offsets run past the message buffer");
    return nullptr;
  }
  napi_value res = make_u8(env, n, &out);
  if (!res) return nullptr;
  if (n && ncg_ed25519_verify_batch_msgs(g_ctx, n, sig, pk, msgs, off.data(), zip215 ? 1 : 0, out) != 0) return throw_native(env);
  return res;
}

// ecdsaVerify(sigs [n*64: r || s big-endian], hashes [n*32], keys [n*33 SEC1 compressed], lowS) -> Uint8Array verdicts
static napi_value EcdsaVerify(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  uint8_t *sig, *hs, *pk, *out;
  size_t sgl, hl, pkl;
  bool low_s = true;
  if (argc < 3 || !get_u8(env, argv[0], &sig, &sgl) || !get_u8(env, argv[1], &hs, &hl) || !get_u8(env, argv[2], &pk, &pkl)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: ecdsaVerify(sigs, hashes, keys, lowS)");
    return nullptr;
  }
  if (argc >= 4) napi_get_value_bool(env, argv[3], &low_s);
  const size_t n = sgl / 64;
  if (sgl % 64 || hl != n * 32 || pkl != n * 33) {
    napi_throw_error(env, nullptr, "arrays of signatures, message hashes and public keys must have matching lengths");
    return nullptr;
  }
  napi_value res = make_u8(env, n, &out);
  if (!res) return nullptr;
  if (n && ncg_ecdsa_verify_batch(g_ctx, NCG_SECP256K1, n, sig, hs, pk, low_s ? NCG_ECDSA_LOW_S : 0, out) != 0) return throw_native(env);
  return res;
}

// secpVerifyMsgs(kind, sigs [n*64], keys [n*33 | n*65 | n*32], msgs, offsets [(n+1)*8 LE], lowS) -> Uint8Array verdicts
// kind 0: ECDSA with prehash (SHA-256 on the device), keys compressed (33 B) or uncompressed (65 B) rows;
// kind 1: BIP-340 Schnorr, x-only keys (32 B), tagged challenge hash on the device
static napi_value SecpVerifyMsgs(napi_env env, napi_callback_info info) {
  size_t argc = 6;
  napi_value argv[6];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  int32_t kind = 0;
  uint8_t *sig, *pk, *msgs, *offs, *out;
  size_t sgl, pkl, ml, ol;
  bool low_s = true;
  if (argc < 5 || napi_get_value_int32(env, argv[0], &kind) != napi_ok || !get_u8(env, argv[1], &sig, &sgl) || !get_u8(env, argv[2], &pk, &pkl) ||
      !get_u8(env, argv[3], &msgs, &ml) || !get_u8(env, argv[4], &offs, &ol)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: secpVerifyMsgs(kind, sigs, keys, msgs, offsets, lowS)");
    return nullptr;
  }
  if (argc >= 6) napi_get_value_bool(env, argv[5], &low_s);
  const size_t n = sgl / 64;
  const size_t kb = n ? pkl / n : 0;
  if (sgl % 64 || ol != (n + 1) * 8 || (n && (pkl % n || (kind == 1 ? kb != 32 : (kb != 33 && kb != 65))))) {
    napi_throw_error(env, nullptr, "arrays of signatures, public keys and message offsets must have matching lengths");
    return nullptr;
  }
  std::vector<uint64_t> off(n + 1);
  memcpy(off.data(), offs, ol);
  if (off[n] > ml) {
    napi_throw_error(env, nullptr, "noble-gpu: message offsets run past the message buffer");
    return nullptr;
  }
  napi_value res = make_u8(env, n, &out);
  if (!res) return nullptr;
  int rc = 0;
  if (n && kind == 0)
    rc = ncg_ecdsa_verify_batch_msgs(g_ctx, NCG_SECP256K1, n, sig, msgs, off.data(), pk,
                                     (low_s ? NCG_ECDSA_LOW_S : 0) | (kb == 65 ? NCG_ECDSA_PUB_UNCOMPRESSED : 0), out);
  else if (n)
    rc = ncg_schnorr_verify_batch_msgs(g_ctx, n, sig, msgs, off.data(), pk, out);
  if (rc != 0) return throw_native(env);
  return res;
}
// ecdsaRecover(sigs65 [n*65: recid || r || s], hashes [n*32]) -> Uint8Array(n*33 compressed keys || n ok flags)
static napi_value EcdsaRecover(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  uint8_t *sig, *hs, *out;
  size_t sgl, hl;
  if (argc < 2 || !get_u8(env, argv[0], &sig, &sgl) || !get_u8(env, argv[1], &hs, &hl)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: ecdsaRecover(sigs65, hashes)");
    return nullptr;
  }
  const size_t n = sgl / 65;
  if (sgl % 65 || hl != n * 32) {
    napi_throw_error(env, nullptr, "arrays of signatures and message hashes must have matching lengths");
    return nullptr;
  }
  napi_value res = make_u8(env, n * 34, &out);
  if (!res) return nullptr;
  if (n) {
    std::vector<uint8_t> aff(n * 64), ok(n), eok(n);
    if (ncg_ecdsa_recover_batch(g_ctx, NCG_SECP256K1, n, sig, hs, aff.data(), ok.data()) != 0) return throw_native(env);
    if (ncg_encode_points_batch(g_ctx, NCG_SECP256K1, n, aff.data(), out, eok.data()) != 0) return throw_native(env);
    for (size_t i = 0; i < n; i++) out[n * 33 + i] = (ok[i] && eok[i]) ? 1 : 0;
  }
  return res;
}

static int encoded_bytes(int curve);
// ---- resident point sets: upload once (affine wire points or compressed encodings), then MSMs / batch
// multiplies with only the scalars crossing.  Handles are small integers; `scalars` may be a Uint8Array
// of packed 32-byte LE values (no marshalling at all) or an Array of BigInt (read with
// napi_get_value_bigint_words - no hex strings).
static std::vector<ncg_points*> g_sets;

static bool get_scalars(napi_env env, napi_value v, size_t n, std::vector<uint8_t>& store, uint8_t** data) {
  size_t len = 0;
  if (get_u8(env, v, data, &len)) return len == n * 32;
  bool is_arr = false;
  if (napi_is_array(env, v, &is_arr) != napi_ok || !is_arr) return false;
  uint32_t m = 0;
  if (napi_get_array_length(env, v, &m) != napi_ok || m != n) return false;
  store.assign(n * 32, 0);
  for (uint32_t i = 0; i < m; i++) {
    napi_value e;
    if (napi_get_element(env, v, i, &e) != napi_ok) return false;
    int sign = 0;
    size_t words = 4;
    uint64_t w[4] = {0, 0, 0, 0};
    if (napi_get_value_bigint_words(env, e, &sign, &words, w) != napi_ok || sign != 0 || words > 4) return false;
    memcpy(store.data() + (size_t)i * 32, w, 32);  // little-endian host
  }
  *data = store.data();
  return true;
}

// packBigInts(values: BigInt[], byteLen) -> Uint8Array(values.length * byteLen), little-endian: the marshalling of
// coordinates / scalars without hex strings (napi_get_value_bigint_words); negative or too wide values throw
static napi_value PackBigInts(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  bool is_arr = false;
  uint32_t m = 0, blen = 0;
  if (argc < 2 || napi_is_array(env, argv[0], &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, argv[0], &m) != napi_ok ||
      napi_get_value_uint32(env, argv[1], &blen) != napi_ok || blen == 0 || blen > 64 || blen % 8) {
    napi_throw_type_error(env, nullptr, "noble-gpu: packBigInts(BigInt[], byteLen multiple of 8, <= 64)");
    return nullptr;
  }
  uint8_t* out;
  napi_value res = make_u8(env, (size_t)m * blen, &out);
  if (!res) return nullptr;
  const size_t maxw = blen / 8;
  napi_handle_scope scope = nullptr;
  for (uint32_t i = 0; i < m; i++) {
    // the element handles of a million-entry array would otherwise all live until the call returns: a fresh
    // handle scope every 4096 elements keeps the handle stack bounded (no speed-up measured: the ~100 ns per value
    // are napi_get_element + napi_get_value_bigint_words themselves)
    if ((i & 4095u) == 0) {
      if (scope) (void)napi_close_handle_scope(env, scope);
      scope = nullptr;
      if (napi_open_handle_scope(env, &scope) != napi_ok) scope = nullptr;
    }
    napi_value e;
    int sign = 0;
    size_t words = 8;
    uint64_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (napi_get_element(env, argv[0], i, &e) != napi_ok || napi_get_value_bigint_words(env, e, &sign, &words, w) != napi_ok) {
      if (scope) (void)napi_close_handle_scope(env, scope);
      napi_throw_type_error(env, nullptr, "noble-gpu: packBigInts: expected bigint");
      return nullptr;
    }
    // `words` comes back as the count the value NEEDS (it may exceed the 8 fetched): a canonical BigInt with more
    // words than the field holds is never zero (e.g. 1n << 512n has eight zero low words), and -0n does not exist
    if (sign != 0 || words > maxw) {
      if (scope) (void)napi_close_handle_scope(env, scope);
      napi_throw_range_error(env, nullptr, "noble-gpu: packBigInts: value out of range");
      return nullptr;
    }
    memcpy(out + (size_t)i * blen, w, blen);  // little-endian host
  }
  if (scope) (void)napi_close_handle_scope(env, scope);
  return res;
}

// unpackBigInts(bytes: Uint8Array, byteLen) -> BigInt[]: little-endian fields of byteLen bytes each
static napi_value UnpackBigInts(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  uint8_t* buf;
  size_t len;
  uint32_t blen = 0;
  if (argc < 2 || !get_u8(env, argv[0], &buf, &len) || napi_get_value_uint32(env, argv[1], &blen) != napi_ok || blen == 0 || blen > 64 ||
      blen % 8 || len % blen) {
    napi_throw_type_error(env, nullptr, "noble-gpu: unpackBigInts(Uint8Array, byteLen multiple of 8, <= 64)");
    return nullptr;
  }
  const size_t m = len / blen;
  napi_value arr;
  NAPI_OK(napi_create_array_with_length(env, m, &arr));
  for (size_t i = 0; i < m; i++) {
    uint64_t w[8];
    memcpy(w, buf + i * blen, blen);
    napi_value v;
    NAPI_OK(napi_create_bigint_words(env, 0, blen / 8, w, &v));
    NAPI_OK(napi_set_element(env, arr, (uint32_t)i, v));
  }
  return arr;
}

static napi_value UploadPoints(napi_env env, napi_callback_info info) {  // (curveId, Uint8Array affine | encoded, encoded?, zip215?)
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  int32_t curve;
  uint8_t* buf;
  size_t len;
  bool encoded = false, zip215 = false;
  if (argc < 2 || napi_get_value_int32(env, argv[0], &curve) != napi_ok || !get_u8(env, argv[1], &buf, &len)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: uploadPoints(curveId, Uint8Array, encoded, zip215)");
    return nullptr;
  }
  if (argc >= 3) napi_get_value_bool(env, argv[2], &encoded);
  if (argc >= 4) napi_get_value_bool(env, argv[3], &zip215);
  const int unit = encoded ? encoded_bytes(curve) : ncg_point_bytes(curve);
  if (unit == 0 || len % unit) {
    napi_throw_error(env, nullptr, "noble-gpu: uploadPoints: buffer is not a whole number of points");
    return nullptr;
  }
  ncg_points* h = nullptr;
  int64_t bad = -1;
  int rc = encoded ? ncg_points_from_encoded(g_ctx, curve, len / unit, buf, zip215 ? 1 : 0, &h, &bad)
                   : ncg_points_upload(g_ctx, curve, len / unit, buf, &h);
  if (rc != 0) return throw_native(env);
  size_t slot = g_sets.size();
  for (size_t i = 0; i < g_sets.size(); i++)
    if (!g_sets[i]) { slot = i; break; }
  if (slot == g_sets.size()) g_sets.push_back(nullptr);
  g_sets[slot] = h;
  napi_value r;
  napi_create_uint32(env, (uint32_t)slot, &r);
  return r;
}

static ncg_points* get_set(napi_env env, napi_value v) {
  uint32_t id = 0;
  if (napi_get_value_uint32(env, v, &id) != napi_ok || id >= g_sets.size() || !g_sets[id]) {
    napi_throw_error(env, nullptr, "noble-gpu: unknown point-set handle");
    return nullptr;
  }
  return g_sets[id];
}

static napi_value FreePoints(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  uint32_t id = 0;
  if (argc >= 1 && napi_get_value_uint32(env, argv[0], &id) == napi_ok && id < g_sets.size() && g_sets[id]) {
    ncg_points_free(g_sets[id]);
    g_sets[id] = nullptr;
  }
  return nullptr;
}

// (handle) -> -1 if every point lies in the prime-order subgroup (the set then takes the endomorphism MSM),
// else the index of the first point outside it (the set keeps the generic path); bls12-381 only
static napi_value VerifySubgroup(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  ncg_points* h = argc >= 1 ? get_set(env, argv[0]) : nullptr;
  if (!h) return nullptr;
  int64_t bad = -1;
  if (ncg_points_verify_subgroup(g_ctx, h, &bad) != NCG_OK) {
    napi_throw_error(env, nullptr, ncg_last_error(g_ctx));
    return nullptr;
  }
  napi_value r;
  NAPI_OK(napi_create_int64(env, bad, &r));
  return r;
}
// precomputePoints(handle) -> boolean: interleavedMSMUnsafe's per-point tables in device form (ncg_points_precompute)
static napi_value PrecomputePoints(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  ncg_points* h = argc >= 1 ? get_set(env, argv[0]) : nullptr;
  if (!h) return nullptr;
  if (ncg_points_precompute(g_ctx, h) != NCG_OK) {
    napi_throw_error(env, nullptr, ncg_last_error(g_ctx));
    return nullptr;
  }
  napi_value r;
  NAPI_OK(napi_get_boolean(env, ncg_points_precomputed(h) != 0, &r));
  return r;
}
static napi_value InSubgroup(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  ncg_points* h = argc >= 1 ? get_set(env, argv[0]) : nullptr;
  if (!h) return nullptr;
  napi_value r;
  NAPI_OK(napi_get_boolean(env, ncg_points_in_subgroup(h) != 0, &r));
  return r;
}

static napi_value MsmResident(napi_env env, napi_callback_info info) {  // (handle, scalars) -> affine || inf
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  ncg_points* h = argc >= 1 ? get_set(env, argv[0]) : nullptr;
  if (!h) return nullptr;
  const size_t n = ncg_points_count(h);
  std::vector<uint8_t> store;
  uint8_t *sc = nullptr, *out;
  if (argc < 2 || !get_scalars(env, argv[1], n, store, &sc)) {
    napi_throw_error(env, nullptr, "arrays of points and scalars must have equal length");
    return nullptr;
  }
  const int pb = ncg_point_bytes(ncg_points_curve(h));
  napi_value res = make_u8(env, pb + 1, &out);
  if (!res) return nullptr;
  if (ncg_msm_resident(g_ctx, h, sc, out, out + pb) != 0) return throw_native(env);
  return res;
}

static napi_value MulVarResident(napi_env env, napi_callback_info info) {  // (handle, scalars) -> n x (affine) || n x inf
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  ncg_points* h = argc >= 1 ? get_set(env, argv[0]) : nullptr;
  if (!h) return nullptr;
  const size_t n = ncg_points_count(h);
  std::vector<uint8_t> store;
  uint8_t *sc = nullptr, *out;
  if (argc < 2 || !get_scalars(env, argv[1], n, store, &sc)) {
    napi_throw_error(env, nullptr, "arrays of points and scalars must have equal length");
    return nullptr;
  }
  const int pb = ncg_point_bytes(ncg_points_curve(h));
  napi_value res = make_u8(env, n * (pb + 1), &out);
  if (!res) return nullptr;
  if (n && ncg_mul_var_batch_resident(g_ctx, h, sc, out, out + n * pb) != 0) return throw_native(env);
  return res;
}

static int encoded_bytes(int curve) { return curve == 0 ? 33 : curve == 1 ? 32 : curve == 2 ? 48 : curve == 3 ? 96 : 0; }

static napi_value DecodePoints(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  int32_t curve;
  uint8_t *enc, *out;
  size_t el;
  bool zip215 = false;
  if (argc < 2 || napi_get_value_int32(env, argv[0], &curve) != napi_ok || !get_u8(env, argv[1], &enc, &el)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: decodePoints(curveId, Uint8Array, zip215)");
    return nullptr;
  }
  if (argc >= 3) napi_get_value_bool(env, argv[2], &zip215);
  int eb = encoded_bytes(curve), pb = ncg_point_bytes(curve);
  if (!eb || el % eb) {
    napi_throw_error(env, nullptr, "noble-gpu: decodePoints: bad curve or length");
    return nullptr;
  }
  size_t n = el / eb;
  napi_value res = make_u8(env, n * (pb + 2), &out);
  if (!res) return nullptr;
  if (n && ncg_decode_points_batch(g_ctx, curve, n, enc, zip215 ? NCG_DECODE_ZIP215 : 0, out, out + n * pb, out + n * pb + n) != 0)
    return throw_native(env);
  return res;
}

static napi_value AggregateEncoded(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  int32_t curve;
  uint8_t *enc, *out;
  size_t el;
  bool zip215 = false;
  if (argc < 2 || napi_get_value_int32(env, argv[0], &curve) != napi_ok || !get_u8(env, argv[1], &enc, &el)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: aggregateEncoded(curveId, Uint8Array, zip215)");
    return nullptr;
  }
  if (argc >= 3) napi_get_value_bool(env, argv[2], &zip215);
  int eb = encoded_bytes(curve), pb = ncg_point_bytes(curve);
  if (!eb || el % eb) {
    napi_throw_error(env, nullptr, "noble-gpu: aggregateEncoded: bad curve or length");
    return nullptr;
  }
  napi_value res = make_u8(env, pb + 1, &out);
  if (!res) return nullptr;
  int64_t bad = -1;
  if (ncg_aggregate_encoded(g_ctx, curve, el / eb, enc, zip215 ? NCG_DECODE_ZIP215 : 0, out, out + pb, &bad) != 0)
    return throw_native(env);
  return res;
}

static napi_value EncodePoints(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  int32_t curve;
  uint8_t *pts, *out;
  size_t pl;
  if (argc < 2 || napi_get_value_int32(env, argv[0], &curve) != napi_ok || !get_u8(env, argv[1], &pts, &pl)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: encodePoints(curveId, Uint8Array)");
    return nullptr;
  }
  int eb = encoded_bytes(curve), pb = ncg_point_bytes(curve);
  if (!eb || pl % pb) {
    napi_throw_error(env, nullptr, "noble-gpu: encodePoints: bad curve or length");
    return nullptr;
  }
  size_t n = pl / pb;
  napi_value res = make_u8(env, n * (eb + 1), &out);
  if (!res) return nullptr;
  if (n && ncg_encode_points_batch(g_ctx, curve, n, pts, out, out + n * eb) != 0) return throw_native(env);
  return res;
}

static napi_value Ntt(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  int32_t log2n, flags = 0;
  uint8_t *om, *data, *out;
  size_t ol, dl;
  if (argc < 3 || napi_get_value_int32(env, argv[0], &log2n) != napi_ok || !get_u8(env, argv[1], &om, &ol) || ol != 32 ||
      !get_u8(env, argv[2], &data, &dl)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: ntt(log2n, omega32, data, flags)");
    return nullptr;
  }
  if (argc >= 4) napi_get_value_int32(env, argv[3], &flags);
  if (log2n < 0 || log2n > NCG_NTT_MAX_LOG2N || dl % ((size_t)32 << log2n)) {
    napi_throw_error(env, nullptr, "FFT: Polynomial size should be power of two");
    return nullptr;
  }
  napi_value res = make_u8(env, dl, &out);
  if (!res) return nullptr;
  size_t batch = dl / ((size_t)32 << log2n);
  if (batch && ncg_ntt(g_ctx, NCG_FIELD_BLS12_381_FR, log2n, batch, om, data, out, flags) != 0) return throw_native(env);
  return res;
}

static napi_value MapToCurve(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!need_ctx(env)) return nullptr;
  int32_t curve, count;
  uint8_t *u, *out;
  size_t ul;
  if (argc < 3 || napi_get_value_int32(env, argv[0], &curve) != napi_ok || napi_get_value_int32(env, argv[1], &count) != napi_ok ||
      !get_u8(env, argv[2], &u, &ul)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: mapToCurve(curveId, count, Uint8Array)");
    return nullptr;
  }
  int pb = ncg_point_bytes(curve);
  size_t per = (size_t)count * (pb / 2);
  if ((curve != NCG_BLS12_381_G1 && curve != NCG_BLS12_381_G2) || (count != 1 && count != 2) || ul % per) {
    napi_throw_error(env, nullptr, "noble-gpu: mapToCurve: bad curve, count or length");
    return nullptr;
  }
  size_t n = ul / per;
  napi_value res = make_u8(env, n * (pb + 1), &out);
  if (!res) return nullptr;
  if (n && ncg_map_to_curve_batch(g_ctx, curve, n, count, u, out, out + n * pb) != 0) return throw_native(env);
  return res;
}

static napi_value PointBytes(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], r;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  int32_t curve = -1;
  if (argc >= 1) napi_get_value_int32(env, argv[0], &curve);
  NAPI_OK(napi_create_int32(env, ncg_point_bytes(curve), &r));
  return r;
}

// hostRegister(Uint8Array) / hostUnregister(Uint8Array): pin a long-lived input / output buffer ONCE (ncg_host_register),
// so that every later call on it moves by DMA at PCIe speed with no per-call page locking
// A registered array must outlive its registration: a strong reference is held from hostRegister until hostUnregister, so the
// backing store cannot be collected (and its address range handed to another allocation) while HIP still has it pinned.
// The map is process-wide (ncg_host_register is) while a napi_ref belongs to ONE env: worker_threads each load the addon, so the
// map is guarded by a mutex and an entry remembers the env that made it; only that env may delete its reference.
struct Registered { napi_env env; napi_ref ref; size_t len; };
static std::map<uintptr_t, Registered> g_registered;
static std::mutex g_registered_mu;
static napi_value HostRegister(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  uint8_t* p;
  size_t len;
  if (argc < 1 || !get_u8(env, argv[0], &p, &len)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: hostRegister(Uint8Array)");
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_registered_mu);
  auto have = g_registered.find((uintptr_t)p);
  if (have != g_registered.end()) {
    if (have->second.len >= len) return nullptr;   // already registered through this addon, at least this long
    // the same base with a LARGER length: silently succeeding would leave the tail unpinned
    napi_throw_error(env, nullptr, "noble-gpu: hostRegister: this buffer is already registered with a smaller length - hostUnregister it first");
    return nullptr;
  }
  if (ncg_host_register(p, len) != 0) return throw_native(env);
  napi_ref ref = nullptr;
  if (napi_create_reference(env, argv[0], 1, &ref) != napi_ok) {
    (void)ncg_host_unregister(p);
    napi_throw_error(env, nullptr, "noble-gpu: hostRegister: cannot hold a reference to the array");
    return nullptr;
  }
  g_registered[(uintptr_t)p] = Registered{env, ref, len};
  return nullptr;
}
static napi_value HostUnregister(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  uint8_t* p;
  size_t len;
  if (argc < 1 || !get_u8(env, argv[0], &p, &len)) {
    napi_throw_type_error(env, nullptr, "noble-gpu: hostUnregister(Uint8Array)");
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_registered_mu);
  auto it = g_registered.find((uintptr_t)p);
  if (it != g_registered.end() && it->second.env != env) {
    napi_throw_error(env, nullptr, "noble-gpu: hostUnregister: this buffer was registered from another thread's environment");
    return nullptr;
  }
  (void)ncg_host_unregister(p);
  if (it != g_registered.end()) {
    (void)napi_delete_reference(env, it->second.ref);
    g_registered.erase(it);
  }
  return nullptr;
}

static napi_value Version(napi_env env, napi_callback_info) {
  napi_value r;
  NAPI_OK(napi_create_string_utf8(env, ncg_version(), NAPI_AUTO_LENGTH, &r));
  return r;
}

NAPI_MODULE_INIT() {
  struct {
    const char* name;
    napi_callback fn;
  } fns[] = {{"init", Init},           {"initMulti", InitMulti}, {"msm", Msm},
             {"mulVarBatch", MulVarBatch}, {"mulBaseBatch", MulBaseBatch},
             {"ed25519VerifyBatch", Ed25519VerifyBatch}, {"pointBytes", PointBytes},
             {"decodePoints", DecodePoints}, {"encodePoints", EncodePoints},
             {"aggregateEncoded", AggregateEncoded},
             {"ntt", Ntt},               {"mapToCurve", MapToCurve},
             {"packBigInts", PackBigInts}, {"unpackBigInts", UnpackBigInts}, {"uploadPoints", UploadPoints}, {"freePoints", FreePoints}, {"verifySubgroup", VerifySubgroup}, {"precomputePoints", PrecomputePoints}, {"inSubgroup", InSubgroup},
             {"msmResident", MsmResident}, {"mulVarResident", MulVarResident},
             {"ed25519VerifyMsgs", Ed25519VerifyMsgs}, {"ecdsaVerify", EcdsaVerify}, {"secpVerifyMsgs", SecpVerifyMsgs}, {"ecdsaRecover", EcdsaRecover},
             {"hostRegister", HostRegister}, {"hostUnregister", HostUnregister},
             {"version", Version}};
  for (auto& f : fns) {
    napi_value v;
    if (napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &v) != napi_ok) return exports;
    napi_set_named_property(env, exports, f.name, v);
  }
  return exports;
}
