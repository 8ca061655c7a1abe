"""ctypes binding of libncg.so (include/ncg.h) plus wire-format marshalling helpers.

The library is built in-tree by `__graft_entry__.build()` (or `make -C noble-curves_amd/csrc`).
It is loaded lazily and LOUDLY: a missing library or a missing GPU raises NativeError - no
silent fallback exists.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

SECP256K1, ED25519, BLS12_381_G1, BLS12_381_G2 = 0, 1, 2, 3
POINT_BYTES = {SECP256K1: 64, ED25519: 64, BLS12_381_G1: 96, BLS12_381_G2: 192}
FIELD_BYTES = {SECP256K1: 32, ED25519: 32, BLS12_381_G1: 48, BLS12_381_G2: 48}
FIELD_BLS12_381_FR = 0
ENCODED_BYTES = {SECP256K1: 33, ED25519: 32, BLS12_381_G1: 48, BLS12_381_G2: 96}   # compressed toBytes


class NativeError(RuntimeError):
    pass


def lib_path():
    """In-tree library; NCG_LIB overrides it (A/B runs of alternative builds)."""
    return os.environ.get("NCG_LIB") or os.path.join(_HERE, "libncg.so")


_lib = None
_lib_lock = threading.Lock()


def load_library():
    """dlopen libncg.so and declare prototypes; raises NativeError if it is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise NativeError(
                "noble-gpu: %s is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % path)
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise NativeError("noble-gpu: cannot load %s: %s" % (path, e))
        vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        lib.ncg_init.argtypes = [i32, ctypes.POINTER(vp)]
        lib.ncg_init.restype = i32
        lib.ncg_destroy.argtypes = [vp]
        lib.ncg_destroy.restype = None
        lib.ncg_last_error.argtypes = [vp]
        lib.ncg_last_error.restype = ctypes.c_char_p
        lib.ncg_sync.argtypes = [vp]
        lib.ncg_sync.restype = i32
        lib.ncg_version.argtypes = []
        lib.ncg_version.restype = ctypes.c_char_p
        lib.ncg_point_bytes.argtypes = [i32]
        lib.ncg_point_bytes.restype = i32
        lib.ncg_field_bytes.argtypes = [i32]
        lib.ncg_field_bytes.restype = i32
        lib.ncg_mul_var_batch.argtypes = [vp, i32, sz, vp, vp, vp, vp]
        lib.ncg_mul_var_batch.restype = i32
        lib.ncg_mul_var_batch_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp, vp]
        lib.ncg_mul_var_batch_dev.restype = i32
        lib.ncg_ubench.argtypes = [vp, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_float)]
        lib.ncg_ubench.restype = i32
        for name, args in _OPTIONAL_PROTOS.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.argtypes = args
                fn.restype = i32
        lib.ncg_points_free.argtypes = [vp]
        lib.ncg_points_free.restype = None
        lib.ncg_points_count.argtypes = [vp]
        lib.ncg_points_count.restype = sz
        lib.ncg_points_dev.argtypes = [vp]
        lib.ncg_points_dev.restype = vp
        lib.ncg_multi_destroy.argtypes = [vp]
        lib.ncg_multi_destroy.restype = None
        lib.ncg_multi_ctx.argtypes = [vp, i32]
        lib.ncg_multi_ctx.restype = vp
        lib.ncg_multi_last_error.argtypes = [vp]
        lib.ncg_multi_last_error.restype = ctypes.c_char_p
        _lib = lib
        return lib


_vp, _sz, _i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
_OPTIONAL_PROTOS = {
    "ncg_decode_points_batch": [_vp, _i32, _sz, _vp, _i32, _vp, _vp, _vp],
    "ncg_decode_points_batch_dev": [_vp, _i32, _sz, _vp, _i32, _vp, _vp, _vp, _vp],
    "ncg_add_pairs_batch": [_vp, _i32, _sz, _vp, _vp, _i32, _vp, _vp],
    "ncg_add_pairs_batch_dev": [_vp, _i32, _sz, _vp, _vp, _i32, _vp, _vp, _vp],
    "ncg_aggregate_encoded": [_vp, _i32, _sz, _vp, _i32, _vp, _vp, _vp],
    "ncg_encode_points_batch": [_vp, _i32, _sz, _vp, _vp, _vp],
    "ncg_encode_points_batch_dev": [_vp, _i32, _sz, _vp, _vp, _vp, _vp],
    "ncg_map_to_curve_batch": [_vp, _i32, _sz, _i32, _vp, _vp, _vp],
    "ncg_map_to_curve_batch_dev": [_vp, _i32, _sz, _i32, _vp, _vp, _vp, _vp],
    "ncg_ntt": [_vp, _i32, _i32, _sz, _vp, _vp, _vp, _i32],
    "ncg_ntt_dev": [_vp, _i32, _i32, _sz, _vp, _vp, _vp, _i32, _vp],
    "ncg_normalize_batch": [_vp, _i32, _sz, _vp, _vp, _vp],
    "ncg_normalize_batch_dev": [_vp, _i32, _sz, _vp, _vp, _vp, _vp],
    "ncg_msm": [_vp, _i32, _sz, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint8)],
    "ncg_msm_dev": [_vp, _i32, _sz, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint8), _vp],
    "ncg_mul_base_batch": [_vp, _i32, _sz, _vp, _vp, _vp],
    "ncg_mul_base_batch_dev": [_vp, _i32, _sz, _vp, _vp, _vp, _vp],
    "ncg_ed25519_verify_batch": [_vp, _sz, _vp, _vp, _vp, _i32, _vp],
    "ncg_ed25519_verify_batch_dev": [_vp, _sz, _vp, _vp, _vp, _i32, _vp, _vp],
    "ncg_ed25519_verify_batch_msgs": [_vp, _sz, _vp, _vp, _vp, _vp, _i32, _vp],
    "ncg_ed25519_verify_batch_msgs_dev": [_vp, _sz, _vp, _vp, _vp, _vp, _i32, _vp, _vp],
    "ncg_ed25519_challenge_batch_dev": [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp],
    "ncg_points_upload": [_vp, _i32, _sz, _vp, ctypes.POINTER(_vp)],
    "ncg_points_from_encoded": [_vp, _i32, _sz, _vp, _i32, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_int64)],
    "ncg_points_curve": [_vp],
    "ncg_msm_resident": [_vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint8)],
    "ncg_msm_resident_dev": [_vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint8), _vp],
    "ncg_ecdsa_verify_batch": [_vp, _i32, _sz, _vp, _vp, _vp, _i32, _vp],
    "ncg_ecdsa_recover_batch": [_vp, _i32, _sz, _vp, _vp, _vp, _vp],
    "ncg_ecdsa_recover_batch_dev": [_vp, _i32, _sz, _vp, _vp, _vp, _vp, _vp],
    "ncg_schnorr_verify_batch": [_vp, _sz, _vp, _vp, _vp, _vp],
    "ncg_ecdsa_verify_batch_msgs": [_vp, _i32, _sz, _vp, _vp, _vp, _vp, _i32, _vp],
    "ncg_ecdsa_verify_batch_msgs_dev": [_vp, _i32, _sz, _vp, _vp, _vp, _vp, _i32, _vp, _vp],
    "ncg_schnorr_verify_batch_msgs": [_vp, _sz, _vp, _vp, _vp, _vp, _vp],
    "ncg_schnorr_verify_batch_msgs_dev": [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp],
    "ncg_schnorr_verify_batch_dev": [_vp, _sz, _vp, _vp, _vp, _vp, _vp],
    "ncg_ecdsa_verify_batch_dev": [_vp, _i32, _sz, _vp, _vp, _vp, _i32, _vp, _vp],
    "ncg_points_verify_subgroup": [_vp, _vp, ctypes.POINTER(ctypes.c_int64)],
    "ncg_points_in_subgroup": [_vp],
    "ncg_points_precompute": [_vp, _vp],
    "ncg_points_precomputed": [_vp],
    "ncg_mul_var_batch_resident": [_vp, _vp, _vp, _vp, _vp],
    "ncg_mul_var_batch_resident_dev": [_vp, _vp, _vp, _vp, _vp, _vp],
    "ncg_field_check": [_vp, _i32, _i32, _i32, _sz, _vp, _vp, _vp],
    "ncg_comm_unique_id": [_vp],
    "ncg_comm_init": [_vp, _i32, _i32, _vp],
    "ncg_comm_destroy": [_vp],
    "ncg_comm_size": [_vp],
    "ncg_comm_rank": [_vp],
    "ncg_comm_count": [_vp, _vp, _vp],
    "ncg_msm_sharded_dev": [_vp, _i32, _sz, _sz, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint8), _vp],
    "ncg_msm_split_dev": [_vp, _i32, _sz, _i32, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint8), _vp],
    "ncg_msm_shard_local_dev": [_vp, _i32, _sz, _sz, _vp, _vp, _vp, _vp],
    "ncg_msm_shard_combine": [_vp, _i32, _sz, _i32, _vp, _vp, ctypes.POINTER(ctypes.c_uint8), _vp],
    "ncg_msm_sharded_windows_dev": [_vp, _i32, _sz, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint8), _vp],
    "ncg_msm_shard_windows_local_dev": [_vp, _i32, _sz, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "ncg_msm_split_windows_dev": [_vp, _i32, _sz, _i32, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint8), _vp],
    "ncg_msm_async_submit": [_vp, _i32, _i32, _sz, _vp, _vp, _vp, _i32, _vp],
    "ncg_msm_async_collect": [_vp, _i32, _vp, ctypes.POINTER(ctypes.c_uint8)],
    "ncg_msm_async_collect_slot": [_vp, _i32, _vp],
    "ncg_msm_set_tuning": [_vp, _i32, _i32],
    "ncg_msm_last_plan": [_vp, ctypes.POINTER(ctypes.c_int)],
    "ncg_multi_init": [ctypes.POINTER(_i32), _i32, ctypes.POINTER(_vp)],
    "ncg_multi_devices": [_vp],
    "ncg_msm_multi": [_vp, _i32, _sz, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint8)],
}
COMM_ID_BYTES = 128


# ---------------------------------------------------------------- wire helpers (numpy, no torch)
def ints_to_le(values, nbytes):
    """list of non-negative ints -> uint8 array [len, nbytes], little-endian."""
    out = np.empty((len(values), nbytes), dtype=np.uint8)
    for i, v in enumerate(values):
        out[i] = np.frombuffer(int(v).to_bytes(nbytes, "little"), dtype=np.uint8)
    return out


def le_to_ints(arr, nbytes):
    """uint8 array [..., nbytes] -> list of ints."""
    flat = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1, nbytes)
    return [int.from_bytes(row.tobytes(), "little") for row in flat]


class Engine:
    """One native context (one GPU).  Methods take/return numpy uint8 arrays in wire format,
    or raw device pointers for the *_dev variants (torch tensors: pass .data_ptr())."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.device = device
        h = ctypes.c_void_p()
        rc = self.lib.ncg_init(device, ctypes.byref(h))
        if rc != 0:
            raise NativeError((self.lib.ncg_last_error(None) or b"").decode() or "noble-gpu: ncg_init failed (%d)" % rc)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.ncg_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = (self.lib.ncg_last_error(self.h) or b"").decode()
            raise NativeError(msg or "noble-gpu: native call failed (%d)" % rc)

    def version(self):
        return self.lib.ncg_version().decode()

    def sync(self):
        self._check(self.lib.ncg_sync(self.h))

    # ---- batch variable-base multiply -------------------------------------------------------
    def mul_var_batch(self, curve, points, scalars, out=None, inf=None):
        """points: uint8 [n, POINT_BYTES]; scalars: uint8 [n, 32] -> (out [n, PB], is_inf [n]).  `out` / `inf`: result arrays the
        caller keeps (and may pin once with host_register) - a fresh 64 MB array per call costs first-touch page faults and a page
        lock before the first result byte lands (2^20 secp256k1 pairs: 43-52 ms per call against 10 with kept, pinned arrays)."""
        pb = POINT_BYTES[curve]
        points = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, pb)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        n = points.shape[0]
        if scalars.shape[0] != n:
            raise ValueError("arrays of points and scalars must have equal length")
        if out is None:
            out = np.empty((n, pb), dtype=np.uint8)
        elif out.dtype != np.uint8 or out.shape != (n, pb) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous uint8 array of shape (n, point bytes)")
        if inf is None:
            inf = np.empty((n,), dtype=np.uint8)
        elif inf.dtype != np.uint8 or inf.shape != (n,) or not inf.flags["C_CONTIGUOUS"]:
            raise ValueError("inf must be a C-contiguous uint8 array of shape (n,)")
        if n:
            self._check(self.lib.ncg_mul_var_batch(self.h, curve, n, points.ctypes.data, scalars.ctypes.data,
                                                   out.ctypes.data, inf.ctypes.data))
        return out, inf

    def decode_points_batch(self, curve, encoded, zip215=False):
        """encoded uint8 [n, 33|32|48|96] -> (affine [n, PB], ok [n] bool, is_inf [n] bool)."""
        enc_bytes = ENCODED_BYTES[curve]
        pb = POINT_BYTES[curve]
        enc = np.ascontiguousarray(encoded, dtype=np.uint8).reshape(-1, enc_bytes)
        n = enc.shape[0]
        out = np.zeros((n, pb), dtype=np.uint8)
        ok = np.zeros((n,), dtype=np.uint8)
        inf = np.zeros((n,), dtype=np.uint8)
        if n:
            self._check(self.lib.ncg_decode_points_batch(self.h, curve, n, enc.ctypes.data, 1 if zip215 else 0,
                                                         out.ctypes.data, ok.ctypes.data, inf.ctypes.data))
        return out, ok.astype(bool), inf.astype(bool)

    def add_pairs_batch(self, curve, a, b, subtract=False):
        """a, b uint8 [n, PB] affine wire points -> (a[i] + b[i] (or a[i] - b[i]) [n, PB], is_inf [n])."""
        pb = POINT_BYTES[curve]
        a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, pb)
        b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, pb)
        if a.shape != b.shape:
            raise ValueError("noble-gpu: add_pairs_batch: operand arrays differ in length")
        out = np.empty_like(a)
        inf = np.empty((a.shape[0],), dtype=np.uint8)
        if a.shape[0]:
            self._check(self.lib.ncg_add_pairs_batch(self.h, curve, a.shape[0], a.ctypes.data, b.ctypes.data,
                                                     1 if subtract else 0, out.ctypes.data, inf.ctypes.data))
        return out, inf

    def aggregate_encoded(self, curve, encoded, zip215=False):
        """encoded uint8 [n, ENCODED_BYTES] -> (affine [PB], is_inf, bad_index): sum of the decoded points;
        bad_index >= 0 (and no result) when an entry does not decode."""
        enc = np.ascontiguousarray(encoded, dtype=np.uint8).reshape(-1, ENCODED_BYTES[curve])
        out = np.zeros((POINT_BYTES[curve],), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        bad = ctypes.c_int64(-1)
        rc = self.lib.ncg_aggregate_encoded(self.h, curve, enc.shape[0], enc.ctypes.data, 1 if zip215 else 0,
                                            out.ctypes.data, ctypes.byref(inf), ctypes.byref(bad))
        if rc and bad.value >= 0:
            return None, False, int(bad.value)
        self._check(rc)
        return out, bool(inf.value), -1

    def encode_points_batch(self, curve, affine):
        """affine uint8 [n, PB] -> (encoded [n, 33|32|48|96], ok [n] bool): compressed Point.toBytes."""
        pb = POINT_BYTES[curve]
        aff = np.ascontiguousarray(affine, dtype=np.uint8).reshape(-1, pb)
        n = aff.shape[0]
        out = np.zeros((n, ENCODED_BYTES[curve]), dtype=np.uint8)
        ok = np.zeros((n,), dtype=np.uint8)
        if n:
            self._check(self.lib.ncg_encode_points_batch(self.h, curve, n, aff.ctypes.data, out.ctypes.data,
                                                         ok.ctypes.data))
        return out, ok.astype(bool)

    def normalize_batch(self, curve, proj):
        """proj uint8 [n, 3*FIELD_BYTES*(2 for Fp2)] (X || Y || Z) -> (affine [n, PB], is_inf [n])."""
        pb = POINT_BYTES[curve]
        proj = np.ascontiguousarray(proj, dtype=np.uint8).reshape(-1, pb // 2 * 3)
        n = proj.shape[0]
        out = np.empty((n, pb), dtype=np.uint8)
        inf = np.empty((n,), dtype=np.uint8)
        if n:
            self._check(self.lib.ncg_normalize_batch(self.h, curve, n, proj.ctypes.data, out.ctypes.data,
                                                     inf.ctypes.data))
        return out, inf

    def mul_base_batch(self, curve, scalars):
        """scalars uint8 [n, 32] -> (out [n, PB], is_inf [n]) with out[i] = scalars[i] * BASE."""
        pb = POINT_BYTES[curve]
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        n = scalars.shape[0]
        out = np.empty((n, pb), dtype=np.uint8)
        inf = np.empty((n,), dtype=np.uint8)
        if n:
            self._check(self.lib.ncg_mul_base_batch(self.h, curve, n, scalars.ctypes.data, out.ctypes.data,
                                                    inf.ctypes.data))
        return out, inf

    def mul_base_batch_dev(self, curve, n, d_scalars, d_out, d_inf, stream=None):
        self._check(self.lib.ncg_mul_base_batch_dev(self.h, curve, n, d_scalars, d_out, d_inf, stream))

    def mul_var_batch_dev(self, curve, n, d_points, d_scalars, d_out, d_inf, stream=None):
        self._check(self.lib.ncg_mul_var_batch_dev(self.h, curve, n, d_points, d_scalars, d_out, d_inf, stream))

    # ---- pinned host buffers ---------------------------------------------------------------------
    def host_register(self, arr):
        """Pin a long-lived numpy array once (ncg_host_register): host-pointer calls on it then run at PCIe speed."""
        fn = self.lib.ncg_host_register
        fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_size_t], ctypes.c_int
        if fn(arr.ctypes.data, arr.nbytes) != 0:
            raise NativeError((self.lib.ncg_last_error(None) or b"").decode() or "noble-gpu: host_register failed")

    def host_unregister(self, arr):
        fn = self.lib.ncg_host_unregister
        fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
        fn(arr.ctypes.data)

    # ---- multi-scalar multiplication -----------------------------------------------------------
    def msm(self, curve, points, scalars):
        """sum_i scalars[i]*points[i]; returns (affine wire bytes [PB], is_inf bool)."""
        pb = POINT_BYTES[curve]
        points = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, pb)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        n = points.shape[0]
        if scalars.shape[0] != n:
            raise ValueError("arrays of points and scalars must have equal length")
        out = np.zeros((pb,), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        self._check(self.lib.ncg_msm(self.h, curve, n, points.ctypes.data if n else None,
                                     scalars.ctypes.data if n else None, out.ctypes.data, ctypes.byref(inf)))
        return out, bool(inf.value)

    def msm_dev(self, curve, n, d_points, d_scalars, stream=None):
        """device-resident inputs (raw pointers); result comes back to the host."""
        pb = POINT_BYTES[curve]
        out = np.zeros((pb,), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        self._check(self.lib.ncg_msm_dev(self.h, curve, n, d_points, d_scalars, out.ctypes.data, ctypes.byref(inf),
                                         stream))
        return out, bool(inf.value)

    # ---- resident point sets (include/ncg.h "resident point sets") -------------------------------
    def upload_points(self, curve, points):
        """points uint8 [n, PB] -> ResidentPoints (device-resident until .free() / garbage collection)."""
        pb = POINT_BYTES[curve]
        points = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, pb)
        h = ctypes.c_void_p()
        self._check(self.lib.ncg_points_upload(self.h, curve, points.shape[0], points.ctypes.data if points.size else None,
                                               ctypes.byref(h)))
        return ResidentPoints(self, h, curve)

    def upload_encoded(self, curve, encoded, zip215=False):
        """encoded uint8 [n, ENCODED_BYTES] -> (ResidentPoints | None, bad_index): decoded and validated on the device."""
        enc = np.ascontiguousarray(encoded, dtype=np.uint8).reshape(-1, ENCODED_BYTES[curve])
        h = ctypes.c_void_p()
        bad = ctypes.c_int64(-1)
        rc = self.lib.ncg_points_from_encoded(self.h, curve, enc.shape[0], enc.ctypes.data if enc.size else None,
                                              1 if zip215 else 0, ctypes.byref(h), ctypes.byref(bad))
        if rc and bad.value >= 0:
            return None, int(bad.value)
        self._check(rc)
        return ResidentPoints(self, h, curve), -1

    # ---- multi-GPU MSM, one process per GPU (include/ncg.h "multi-GPU MSM" (1)) ------------------
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL unique id (rank 0 creates it and ships it to the other ranks)."""
        lib = load_library()
        buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
        rc = lib.ncg_comm_unique_id(buf)
        if rc != 0:
            raise NativeError((lib.ncg_last_error(None) or b"").decode() or "noble-gpu: ncg_comm_unique_id failed")
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id):
        """Collective: joins this context to the communicator named by `unique_id`."""
        uid = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._check(self.lib.ncg_comm_init(self.h, nranks, rank, uid))
        self._has_comm = True

    def comm_size(self):
        return self.lib.ncg_comm_size(self.h)

    def comm_count(self):
        """(ranks, my rank) as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank); (0, -1) without one."""
        n, me = ctypes.c_int(0), ctypes.c_int(-1)
        self._check(self.lib.ncg_comm_count(self.h, ctypes.byref(n), ctypes.byref(me)))
        return int(n.value), int(me.value)

    def has_comm(self):
        """True between comm_init and comm_destroy (comm_size() reads 1 both without a communicator and with one of a single rank)."""
        return bool(getattr(self, "_has_comm", False))

    def comm_destroy(self):
        self._has_comm = False
        self._check(self.lib.ncg_comm_destroy(self.h))

    def msm_sharded_dev(self, curve, n_local, d_points, d_scalars, stream=None, n_max=0):
        """Collective MSM over the union of all ranks' device-resident shards (RCCL all-gather of the
        grouped window sums + on-device add); every rank gets (affine wire bytes [PB], is_inf)."""
        pb = POINT_BYTES[curve]
        out = np.zeros((pb,), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        self._check(self.lib.ncg_msm_sharded_dev(self.h, curve, n_local, n_max, d_points, d_scalars, out.ctypes.data,
                                                 ctypes.byref(inf), stream))
        return out, bool(inf.value)

    def msm_plan_info(self, curve, n):
        """{c, nwin, nb, ngroups} of the window plan for an n-point MSM (ncg_msm_plan_info)."""
        out = (ctypes.c_int * 4)()
        fn = self.lib.ncg_msm_plan_info
        fn.argtypes, fn.restype = [ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)], ctypes.c_int
        if fn(curve, n, out) != 0:
            raise NativeError("noble-gpu: msm_plan_info: bad arguments")
        return {"c": out[0], "nwin": out[1], "nb": out[2], "ngroups": out[3]}

    def msm_shard_slot_bytes(self, curve):
        fn = self.lib.ncg_msm_shard_slot_bytes
        fn.argtypes, fn.restype = [ctypes.c_int], ctypes.c_size_t
        return int(fn(curve))

    def msm_shard_local_dev(self, curve, n_local, d_points, d_scalars, stream=None, n_max=0):
        """This rank's part of a sharded MSM with a host-staged exchange: per-shard phase on the device, the slot
        (window-plan header + grouped window sums, fixed size per curve) comes back as a uint8 array."""
        slot = np.zeros((self.msm_shard_slot_bytes(curve),), dtype=np.uint8)
        self._check(self.lib.ncg_msm_shard_local_dev(self.h, curve, n_local, n_max, d_points, d_scalars, slot.ctypes.data, stream))
        return slot

    def msm_shard_combine(self, curve, n_max, slots, stream=None):
        """slots: uint8 [nparts, slot_bytes] in rank order -> (affine wire bytes [PB], is_inf): upload, header check,
        adding kernel, finish - what ncg_msm_sharded_dev runs after its all-gather."""
        slots = np.ascontiguousarray(slots, dtype=np.uint8)
        pb = POINT_BYTES[curve]
        out = np.zeros((pb,), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        self._check(self.lib.ncg_msm_shard_combine(self.h, curve, n_max, int(slots.shape[0]), slots.ctypes.data, out.ctypes.data,
                                                   ctypes.byref(inf), stream))
        return out, bool(inf.value)

    def msm_split_dev(self, curve, n, parts, d_points, d_scalars, stream=None):
        """The sharded pipeline on this one GPU: `parts` slices, per-shard phase each, multi-GPU combine
        kernel, finish (ncg_msm_split_dev)."""
        pb = POINT_BYTES[curve]
        out = np.zeros((pb,), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        self._check(self.lib.ncg_msm_split_dev(self.h, curve, n, parts, d_points, d_scalars, out.ctypes.data,
                                               ctypes.byref(inf), stream))
        return out, bool(inf.value)

    # ---- window-sharded MSM (include/ncg.h "WINDOW-sharded mode"): every rank holds all points and scalars ----
    @staticmethod
    def _res_handle(resident):
        return resident.h if resident is not None else None

    def msm_sharded_windows_dev(self, curve, n, d_points, d_scalars, stream=None, resident=None):
        """Collective: rank r runs its range of the windows of ONE n-point MSM; the slots are concatenated (precomputed
        sets: added).  `resident`: a ResidentPoints of this engine instead of d_points."""
        pb = POINT_BYTES[resident.curve if resident is not None else curve]
        out = np.zeros((pb,), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        self._check(self.lib.ncg_msm_sharded_windows_dev(self.h, curve, n, d_points, self._res_handle(resident), d_scalars,
                                                         out.ctypes.data, ctypes.byref(inf), stream))
        return out, bool(inf.value)

    def msm_shard_windows_local_dev(self, curve, n, part, nparts, d_points, d_scalars, stream=None, resident=None):
        """Part `part` of `nparts` of a window-sharded MSM with a host-staged exchange: the slot as a uint8 array."""
        c = resident.curve if resident is not None else curve
        slot = np.zeros((self.msm_shard_slot_bytes(c),), dtype=np.uint8)
        self._check(self.lib.ncg_msm_shard_windows_local_dev(self.h, curve, n, part, nparts, d_points, self._res_handle(resident),
                                                             d_scalars, slot.ctypes.data, stream))
        return slot

    def msm_split_windows_dev(self, curve, n, parts, d_points, d_scalars, stream=None, resident=None):
        """The window-sharded pipeline on this one GPU: the parts in turn, then the concatenation / finish."""
        pb = POINT_BYTES[resident.curve if resident is not None else curve]
        out = np.zeros((pb,), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        self._check(self.lib.ncg_msm_split_windows_dev(self.h, curve, n, parts, d_points, self._res_handle(resident), d_scalars,
                                                       out.ctypes.data, ctypes.byref(inf), stream))
        return out, bool(inf.value)

    # ---- several MSMs in flight (include/ncg.h "Several MSMs in flight") --------------------------------------
    ASYNC_WINDOWS = 1

    def msm_async_lanes(self):
        fn = self.lib.ncg_msm_async_lanes
        fn.argtypes, fn.restype = [], ctypes.c_int
        return int(fn())

    def msm_async_submit(self, lane, curve, n, d_points, d_scalars, stream=None, resident=None, flags=0):
        self._check(self.lib.ncg_msm_async_submit(self.h, lane, curve, n, d_points, self._res_handle(resident), d_scalars, flags, stream))

    def msm_async_collect(self, lane, curve):
        pb = POINT_BYTES[curve]
        out = np.zeros((pb,), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        self._check(self.lib.ncg_msm_async_collect(self.h, lane, out.ctypes.data, ctypes.byref(inf)))
        return out, bool(inf.value)

    @staticmethod
    def async_part(part, nparts):
        """flags for msm_async_submit: ONE part of a window-sharded MSM (NCG_MSM_ASYNC_PART)."""
        return 2 | (part << 8) | (nparts << 20)

    def msm_async_collect_slot(self, lane, curve):
        slot = np.zeros((self.msm_shard_slot_bytes(curve),), dtype=np.uint8)
        self._check(self.lib.ncg_msm_async_collect_slot(self.h, lane, slot.ctypes.data))
        return slot

    def msm_set_tuning(self, seg=0, run_serial=-1):
        """Diagnostics: entries per accumulate lane / serial pieces per cut bucket (0 / -1 = defaults)."""
        self._check(self.lib.ncg_msm_set_tuning(self.h, seg, run_serial))

    def msm_last_plan(self):
        out = (ctypes.c_int * 8)()
        self._check(self.lib.ncg_msm_last_plan(self.h, out))
        keys = ("c", "nwin", "nb", "w0", "nwin_total", "seg", "run_serial", "long_runs")
        return dict(zip(keys, [int(v) for v in out]))

    # ---- ed25519 batch verify --------------------------------------------------------------------
    def ed25519_verify_batch(self, sigs, pks, ks, zip215=True):
        """sigs uint8 [n,64], pks [n,32], ks [n,32] (challenge scalars, LE) -> bool array [n]."""
        sigs = np.ascontiguousarray(sigs, dtype=np.uint8).reshape(-1, 64)
        pks = np.ascontiguousarray(pks, dtype=np.uint8).reshape(-1, 32)
        ks = np.ascontiguousarray(ks, dtype=np.uint8).reshape(-1, 32)
        n = sigs.shape[0]
        if pks.shape[0] != n or ks.shape[0] != n:
            raise ValueError("arrays of signatures, public keys and challenges must have equal length")
        ok = np.zeros((n,), dtype=np.uint8)
        if n:
            self._check(self.lib.ncg_ed25519_verify_batch(self.h, n, sigs.ctypes.data, pks.ctypes.data,
                                                          ks.ctypes.data, 1 if zip215 else 0, ok.ctypes.data))
        return ok.astype(bool)

    def ed25519_verify_batch_msgs(self, sigs, pks, msgs_blob, msg_off, zip215=True):
        """sigs uint8 [n,64], pks [n,32], msgs_blob uint8 [total], msg_off uint64 [n+1] -> bool array [n];
        the challenge hash SHA-512(R || A || M) mod L runs on the device."""
        sigs = np.ascontiguousarray(sigs, dtype=np.uint8).reshape(-1, 64)
        pks = np.ascontiguousarray(pks, dtype=np.uint8).reshape(-1, 32)
        blob = np.ascontiguousarray(msgs_blob, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(msg_off, dtype=np.uint64).reshape(-1)
        n = sigs.shape[0]
        if pks.shape[0] != n or off.shape[0] != n + 1:
            raise ValueError("arrays of signatures, public keys and message offsets must have matching lengths")
        ok = np.zeros((n,), dtype=np.uint8)
        if n:
            self._check(self.lib.ncg_ed25519_verify_batch_msgs(self.h, n, sigs.ctypes.data, pks.ctypes.data,
                                                               blob.ctypes.data if blob.size else None, off.ctypes.data,
                                                               1 if zip215 else 0, ok.ctypes.data))
        return ok.astype(bool)

    def ed25519_verify_batch_msgs_dev(self, n, d_sigs, d_pks, d_msgs, d_off, zip215, d_ok, stream=None):
        self._check(self.lib.ncg_ed25519_verify_batch_msgs_dev(self.h, n, d_sigs, d_pks, d_msgs, d_off,
                                                               1 if zip215 else 0, d_ok, stream))

    def ed25519_challenge_batch_dev(self, n, d_sigs, d_pks, d_msgs, d_off, d_ks, stream=None):
        self._check(self.lib.ncg_ed25519_challenge_batch_dev(self.h, n, d_sigs, d_pks, d_msgs, d_off, d_ks, stream))

    def ed25519_verify_batch_dev(self, n, d_sigs, d_pks, d_ks, zip215, d_ok, stream=None):
        self._check(self.lib.ncg_ed25519_verify_batch_dev(self.h, n, d_sigs, d_pks, d_ks, 1 if zip215 else 0,
                                                          d_ok, stream))

    def ecdsa_verify_batch(self, sigs, hashes, pubs, low_s=True):
        """secp256k1: sigs uint8 [n,64] (r || s big-endian), hashes [n,32], pubs [n,33] SEC1 compressed or [n,65]
        uncompressed -> bool [n] (ecdsa.verify with prehash: false, format 'compact'; weierstrass.ts:1571-1620)."""
        sigs = np.ascontiguousarray(sigs, dtype=np.uint8).reshape(-1, 64)
        hashes = np.ascontiguousarray(hashes, dtype=np.uint8).reshape(-1, 32)
        pubs = np.ascontiguousarray(pubs, dtype=np.uint8)
        n = sigs.shape[0]
        kb = 65 if (pubs.ndim == 2 and pubs.shape[1] == 65) else 33
        pubs = pubs.reshape(-1, kb)
        if hashes.shape[0] != n or pubs.shape[0] != n:
            raise ValueError("arrays of signatures, message hashes and public keys must have equal length")
        ok = np.zeros((n,), dtype=np.uint8)
        if n:
            flags = (1 if low_s else 0) | (2 if kb == 65 else 0)
            self._check(self.lib.ncg_ecdsa_verify_batch(self.h, SECP256K1, n, sigs.ctypes.data, hashes.ctypes.data,
                                                        pubs.ctypes.data, flags, ok.ctypes.data))
        return ok.astype(bool)

    def ecdsa_recover_batch(self, sigs65, hashes):
        """secp256k1: sigs65 uint8 [n,65] (recid || r || s), hashes [n,32] -> (affine points [n,64], ok [n] bool):
        Signature.recoverPublicKey (weierstrass.ts:1391-1407)."""
        sigs = np.ascontiguousarray(sigs65, dtype=np.uint8).reshape(-1, 65)
        hashes = np.ascontiguousarray(hashes, dtype=np.uint8).reshape(-1, 32)
        n = sigs.shape[0]
        if hashes.shape[0] != n:
            raise ValueError("arrays of signatures and message hashes must have equal length")
        out = np.zeros((n, 64), dtype=np.uint8)
        ok = np.zeros((n,), dtype=np.uint8)
        if n:
            self._check(self.lib.ncg_ecdsa_recover_batch(self.h, SECP256K1, n, sigs.ctypes.data, hashes.ctypes.data,
                                                         out.ctypes.data, ok.ctypes.data))
        return out, ok.astype(bool)

    @staticmethod
    def _msg_blob(msgs):
        off = np.zeros((len(msgs) + 1,), np.uint64)
        if len(msgs):
            off[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
        blob = np.frombuffer(b"".join(bytes(m) for m in msgs), np.uint8) if len(msgs) and off[-1] else np.zeros((0,), np.uint8)
        return blob, off

    def ecdsa_verify_batch_msgs(self, sigs, msgs, pubs, low_s=True):
        """The same as ecdsa_verify_batch from the messages themselves: SHA-256 (the reference's prehash) on the device."""
        sigs = np.ascontiguousarray(sigs, dtype=np.uint8).reshape(-1, 64)
        pubs = np.ascontiguousarray(pubs, dtype=np.uint8)
        n = sigs.shape[0]
        kb = 65 if (pubs.ndim == 2 and pubs.shape[1] == 65) else 33
        pubs = pubs.reshape(-1, kb)
        if len(msgs) != n or pubs.shape[0] != n:
            raise ValueError("arrays of signatures, messages and public keys must have equal length")
        blob, off = self._msg_blob(msgs)
        ok = np.zeros((n,), dtype=np.uint8)
        if n:
            flags = (1 if low_s else 0) | (2 if kb == 65 else 0)
            self._check(self.lib.ncg_ecdsa_verify_batch_msgs(self.h, SECP256K1, n, sigs.ctypes.data, blob.ctypes.data if blob.size else None,
                                                             off.ctypes.data, pubs.ctypes.data, flags, ok.ctypes.data))
        return ok.astype(bool)

    def schnorr_verify_batch_msgs(self, sigs, msgs, pubs):
        """BIP-340 verification from the messages: the tagged challenge hash runs on the device too."""
        sigs = np.ascontiguousarray(sigs, dtype=np.uint8).reshape(-1, 64)
        pubs = np.ascontiguousarray(pubs, dtype=np.uint8).reshape(-1, 32)
        n = sigs.shape[0]
        if len(msgs) != n or pubs.shape[0] != n:
            raise ValueError("arrays of signatures, messages and public keys must have equal length")
        blob, off = self._msg_blob(msgs)
        ok = np.zeros((n,), dtype=np.uint8)
        if n:
            self._check(self.lib.ncg_schnorr_verify_batch_msgs(self.h, n, sigs.ctypes.data, blob.ctypes.data if blob.size else None,
                                                               off.ctypes.data, pubs.ctypes.data, ok.ctypes.data))
        return ok.astype(bool)

    def schnorr_verify_batch(self, sigs, challenges, pubs):
        """BIP-340: sigs uint8 [n,64], challenges [n,32] (e mod n, big-endian), pubs [n,32] x-only -> bool [n]
        (src/secp256k1.ts:228-258 after the host-side tagged hash)."""
        sigs = np.ascontiguousarray(sigs, dtype=np.uint8).reshape(-1, 64)
        es = np.ascontiguousarray(challenges, dtype=np.uint8).reshape(-1, 32)
        pubs = np.ascontiguousarray(pubs, dtype=np.uint8).reshape(-1, 32)
        n = sigs.shape[0]
        if es.shape[0] != n or pubs.shape[0] != n:
            raise ValueError("arrays of signatures, challenges and public keys must have equal length")
        ok = np.zeros((n,), dtype=np.uint8)
        if n:
            self._check(self.lib.ncg_schnorr_verify_batch(self.h, n, sigs.ctypes.data, es.ctypes.data, pubs.ctypes.data,
                                                          ok.ctypes.data))
        return ok.astype(bool)

    def schnorr_verify_batch_dev(self, n, d_sigs, d_es, d_pubs, d_ok, stream=None):
        self._check(self.lib.ncg_schnorr_verify_batch_dev(self.h, n, d_sigs, d_es, d_pubs, d_ok, stream))

    def ecdsa_verify_batch_dev(self, n, d_sigs, d_hashes, d_pubs, low_s, d_ok, stream=None, uncompressed=False):
        self._check(self.lib.ncg_ecdsa_verify_batch_dev(self.h, SECP256K1, n, d_sigs, d_hashes, d_pubs,
                                                        (1 if low_s else 0) | (2 if uncompressed else 0), d_ok, stream))

    def map_to_curve_batch(self, curve, u, count):
        """u uint8 [n, count * FIELD_BYTES * (2 for G2)] -> (affine [n, PB], is_inf [n]):
        clearCofactor(sum of the `count` mapped points) per row (bls12-381 G1 / G2)."""
        pb = POINT_BYTES[curve]
        u = np.ascontiguousarray(u, dtype=np.uint8).reshape(-1, count * pb // 2)
        n = u.shape[0]
        out = np.empty((n, pb), dtype=np.uint8)
        inf = np.empty((n,), dtype=np.uint8)
        if n:
            self._check(self.lib.ncg_map_to_curve_batch(self.h, curve, n, count, u.ctypes.data, out.ctypes.data,
                                                        inf.ctypes.data))
        return out, inf

    def map_to_curve_batch_dev(self, curve, n, count, d_u, d_out, d_inf, stream):
        self._check(self.lib.ncg_map_to_curve_batch_dev(self.h, curve, n, count, d_u, d_out, d_inf, stream))

    @staticmethod
    def _ntt_flags(inverse, brp_input, brp_output):
        return (1 if inverse else 0) | (2 if brp_input else 0) | (4 if brp_output else 0)

    def ntt(self, log2n, data, omega, inverse=False, brp_input=False, brp_output=False):
        """data uint8 [batch * 2^log2n, 32] (canonical LE residues of bls12-381 Fr) -> transformed
        copy; omega: int, the primitive 2^log2n-th root of unity (roots.omega(log2n))."""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, 32)
        n = 1 << log2n
        if data.shape[0] % n:
            raise ValueError("noble-gpu: ntt: data is not a whole number of 2^%d-point polynomials" % log2n)
        out = np.empty_like(data)
        om = np.frombuffer(int(omega).to_bytes(32, "little"), dtype=np.uint8).copy()
        if data.shape[0]:
            self._check(self.lib.ncg_ntt(self.h, FIELD_BLS12_381_FR, log2n, data.shape[0] // n, om.ctypes.data,
                                         data.ctypes.data, out.ctypes.data,
                                         self._ntt_flags(inverse, brp_input, brp_output)))
        return out

    def ntt_dev(self, log2n, batch, omega, d_in, d_out, stream, inverse=False, brp_input=False, brp_output=False):
        om = np.frombuffer(int(omega).to_bytes(32, "little"), dtype=np.uint8).copy()
        self._check(self.lib.ncg_ntt_dev(self.h, FIELD_BLS12_381_FR, log2n, batch, om.ctypes.data, d_in, d_out,
                                         self._ntt_flags(inverse, brp_input, brp_output), stream))

    def field_check(self, field, op, variant, a_words, b_words):
        """Device field code on raw operands (ncg_field_check): a_words, b_words uint32 [n, 9] (fields 0/1),
        [n, 12] (field 2), [n, 28] (field 3: raw Fe29 limbs [a, c]) or [n, 56] (field 4: lane-paired Fp2 raw limbs)
        -> uint32 [n, 8 | 12 | 24]."""
        a = np.ascontiguousarray(a_words, dtype=np.uint32)
        b = np.ascontiguousarray(b_words, dtype=np.uint32)
        n = a.shape[0]
        out = np.zeros((n, 24 if field == 4 else 12 if field >= 2 else 8), dtype=np.uint32)
        if n:
            self._check(self.lib.ncg_field_check(self.h, field, op, variant, n, a.ctypes.data, b.ctypes.data, out.ctypes.data))
        return out

    def ubench(self, kind, blocks, threads, iters):
        ms = ctypes.c_float()
        self._check(self.lib.ncg_ubench(self.h, kind, blocks, threads, iters, ctypes.byref(ms)))
        return ms.value


class ResidentPoints:
    """A point set kept in device memory (ncg_points): run MSMs / batch multiplies against it with only
    the scalars crossing."""

    def __init__(self, engine, handle, curve):
        self.engine, self.h, self.curve = engine, handle, curve

    def __len__(self):
        return int(self.engine.lib.ncg_points_count(self.h)) if self.h else 0

    def dev_ptr(self):
        return self.engine.lib.ncg_points_dev(self.h)

    def verify_subgroup(self):
        """bls12-381 G1 / G2: run the reference's subgroup test on every point once (ncg_points_verify_subgroup);
        if all pass, MSMs on this set take the endomorphism path.  Returns -1, or the index of the first point
        outside the prime-order subgroup (the set then keeps the generic path)."""
        bad = ctypes.c_int64(-1)
        self.engine._check(self.engine.lib.ncg_points_verify_subgroup(self.engine.h, self.h, ctypes.byref(bad)))
        return int(bad.value)

    def precompute(self):
        """Build the window-shifted copies of the set once (ncg_points_precompute: the device form of
        interleavedMSMUnsafe's per-point tables, curve.ts:907-959); later MSMs add every window into one bucket
        set.  Returns True if the shared-bucket path is active afterwards (sets of >= 4096 Weierstrass points)."""
        self.engine._check(self.engine.lib.ncg_points_precompute(self.engine.h, self.h))
        return self.precomputed

    @property
    def precomputed(self):
        return bool(self.engine.lib.ncg_points_precomputed(self.h)) if self.h else False

    @property
    def in_subgroup(self):
        """True once the set is known to lie in the prime-order subgroup (decoded, or verified)."""
        return bool(self.engine.lib.ncg_points_in_subgroup(self.h)) if self.h else False

    def free(self):
        if getattr(self, "h", None):
            self.engine.lib.ncg_points_free(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            if getattr(self.engine, "h", None):
                self.free()
        except Exception:
            pass

    def msm(self, scalars):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        if scalars.shape[0] != len(self):
            raise ValueError("arrays of points and scalars must have equal length")
        out = np.zeros((POINT_BYTES[self.curve],), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        self.engine._check(self.engine.lib.ncg_msm_resident(self.engine.h, self.h, scalars.ctypes.data if scalars.size else None,
                                                            out.ctypes.data, ctypes.byref(inf)))
        return out, bool(inf.value)

    def msm_dev(self, d_scalars, stream=None):
        """The same with the scalars already on the device (raw pointer, 32 B LE each, len(self) of them)."""
        out = np.zeros((POINT_BYTES[self.curve],), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        self.engine._check(self.engine.lib.ncg_msm_resident_dev(self.engine.h, self.h, d_scalars, out.ctypes.data,
                                                                ctypes.byref(inf), stream))
        return out, bool(inf.value)

    def mul_var_batch(self, scalars):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        n = len(self)
        if scalars.shape[0] != n:
            raise ValueError("arrays of points and scalars must have equal length")
        out = np.empty((n, POINT_BYTES[self.curve]), dtype=np.uint8)
        inf = np.empty((n,), dtype=np.uint8)
        if n:
            self.engine._check(self.engine.lib.ncg_mul_var_batch_resident(self.engine.h, self.h, scalars.ctypes.data,
                                                                          out.ctypes.data, inf.ctypes.data))
        return out, inf

    def mul_var_batch_dev(self, d_scalars, d_out, d_inf, stream=None):
        """The same with scalars, results and infinity flags in device memory (raw pointers)."""
        self.engine._check(self.engine.lib.ncg_mul_var_batch_resident_dev(self.engine.h, self.h, d_scalars, d_out, d_inf, stream))


class MultiEngine:
    """One process driving several GPUs (include/ncg.h "multi-GPU MSM" (2)): a context per device and one
    RCCL communicator over the set; `msm` shards host arrays over the devices."""

    def __init__(self, device_ids):
        self.lib = load_library()
        ids = (ctypes.c_int * len(device_ids))(*device_ids)
        h = ctypes.c_void_p()
        rc = self.lib.ncg_multi_init(ids, len(device_ids), ctypes.byref(h))
        if rc != 0:
            raise NativeError((self.lib.ncg_last_error(None) or b"").decode() or "noble-gpu: ncg_multi_init failed (%d)" % rc)
        self.h = h

    def devices(self):
        return self.lib.ncg_multi_devices(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ncg_multi_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def msm(self, curve, points, scalars):
        pb = POINT_BYTES[curve]
        points = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, pb)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        n = points.shape[0]
        if scalars.shape[0] != n:
            raise ValueError("arrays of points and scalars must have equal length")
        out = np.zeros((pb,), dtype=np.uint8)
        inf = ctypes.c_uint8(0)
        rc = self.lib.ncg_msm_multi(self.h, curve, n, points.ctypes.data if n else None,
                                    scalars.ctypes.data if n else None, out.ctypes.data, ctypes.byref(inf))
        if rc != 0:
            raise NativeError((self.lib.ncg_multi_last_error(self.h) or b"").decode() or "noble-gpu: msm_multi failed")
        return out, bool(inf.value)


_engines = {}


def get_engine(device=None):
    """Process-wide engine for `device` (default: LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    eng = _engines.get(device)
    if eng is None:
        eng = Engine(device)
        _engines[device] = eng
    return eng
