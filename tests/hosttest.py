"""Loader for the host-only unit-test shim (tests/_build/libncg_hosttest.so), which executes
the kernels' shared __host__ __device__ templates on the CPU.  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "_build", "libncg_hosttest.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        csrc = os.path.join(ROOT, "noble-curves_amd", "csrc")
        subprocess.check_call(["make", "-C", csrc, "hosttest"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _lib = ctypes.CDLL(SO)
        vp, i32 = ctypes.c_void_p, ctypes.c_int
        _lib.ht_mul_var.argtypes = [i32, vp, vp, vp, vp, i32]
        _lib.ht_field_op.argtypes = [i32, i32, vp, vp, vp]
        _lib.ht_glv_split.argtypes = [vp, vp]
        _lib.ht_ed25519_verify.argtypes = [vp, vp, vp, i32]
        _lib.ht_ed25519_mul_var.argtypes = [vp, vp, vp, vp, i32]
        _lib.ht_decode_points.argtypes = [i32, vp, i32, vp, vp, vp, i32]
        _lib.ht_fp2_sqrt.argtypes = [vp, vp]
        _lib.ht_encode_points.argtypes = [i32, vp, vp, vp, i32]
        _lib.ht_map_to_curve.argtypes = [i32, vp, i32, vp, vp, i32]
        _lib.ht_ntt.argtypes = [i32, vp, vp, vp, i32]
        _lib.ht_ntt_plan.argtypes = [i32, vp]
        _lib.ht_ntt_small_passes.argtypes = [i32, vp, vp, vp, i32, i32, i32]
        _lib.ht_fr29_op.argtypes = [i32, vp, vp, vp]
        _lib.ht_fe9_op.argtypes = [i32, i32, i32, vp, vp, vp]
        _lib.ht_ed25519_challenge.argtypes = [vp, vp, vp, ctypes.c_uint64, vp]
        _lib.ht_bls_endo_split.argtypes = [i32, vp, vp]
        _lib.ht_ecdsa_prepare.argtypes = [vp, vp, i32, vp, vp]
        _lib.ht_msm_shard_slot_bytes.argtypes = [i32]
        _lib.ht_msm_shard_slot_bytes.restype = ctypes.c_size_t
        _lib.ht_msm_shard_local.argtypes = [i32, i32, i32, vp, vp, vp]
        _lib.ht_msm_shard_combine.argtypes = [i32, i32, i32, vp, vp, vp, vp, i32]
        _lib.ht_msm_shard_windows_local.argtypes = [i32, i32, i32, i32, vp, vp, vp, i32]
        _lib.ht_msm_finish.argtypes = [i32, i32, i32, vp, vp, vp, i32]
        _lib.ht_msm_plan.argtypes = [i32, i32, vp]
        _lib.ht_msm_seg.argtypes = [i32, i32, i32, vp]
        _lib.ht_msm_plan_top.argtypes = [i32, i32, i32, vp]
        _lib.ht_h64_op.argtypes = [i32, vp, vp, vp, vp]
    return _lib


def mul_var(curve, pts, scalars):
    pts = np.ascontiguousarray(pts, dtype=np.uint8)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
    n = pts.shape[0]
    out = np.zeros_like(pts)
    inf = np.zeros((n,), dtype=np.uint8)
    rc = lib().ht_mul_var(curve, pts.ctypes.data, scalars.ctypes.data, out.ctypes.data, inf.ctypes.data, n)
    assert rc == 0
    return out, inf


def field_op(field, op, a, b, nbytes):
    A = np.frombuffer(int(a).to_bytes(nbytes, "little"), dtype=np.uint8).copy()
    B = np.frombuffer(int(b).to_bytes(nbytes, "little"), dtype=np.uint8).copy()
    R = np.zeros(nbytes, dtype=np.uint8)
    assert lib().ht_field_op(field, op, A.ctypes.data, B.ctypes.data, R.ctypes.data) == 0
    return int.from_bytes(R.tobytes(), "little")


def glv_split(k):
    K = np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).copy()
    out = np.zeros(12, dtype=np.uint32)
    assert lib().ht_glv_split(K.ctypes.data, out.ctypes.data) == 0
    k1 = sum(int(out[i]) << (32 * i) for i in range(5))
    k2 = sum(int(out[5 + i]) << (32 * i) for i in range(5))
    return bool(out[10]), k1, bool(out[11]), k2


def ed_halve(k, s):
    """(u, v, w) of the halved ed25519 verification: u signed, u k == v (mod L), w = |u| s mod L."""
    K = np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).copy()
    S = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8).copy()
    out = np.zeros(17, dtype=np.uint32)
    f = lib().ht_ed_halve
    f.argtypes = [ctypes.c_void_p] * 3
    assert f(K.ctypes.data, S.ctypes.data, out.ctypes.data) == 0
    u = sum(int(out[i]) << (32 * i) for i in range(4))
    v = sum(int(out[4 + i]) << (32 * i) for i in range(4))
    w = sum(int(out[9 + i]) << (32 * i) for i in range(8))
    return (-u if out[8] else u), v, w


def ed25519_verify(sig, pk, k, zip215):
    S = np.frombuffer(bytes(sig), dtype=np.uint8).copy()
    P = np.frombuffer(bytes(pk), dtype=np.uint8).copy()
    K = np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).copy()
    return bool(lib().ht_ed25519_verify(S.ctypes.data, P.ctypes.data, K.ctypes.data, 1 if zip215 else 0))


def ed25519_mul_var(pts, scalars):
    pts = np.ascontiguousarray(pts, dtype=np.uint8)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
    n = pts.shape[0]
    out = np.zeros_like(pts)
    inf = np.zeros((n,), dtype=np.uint8)
    assert lib().ht_ed25519_mul_var(pts.ctypes.data, scalars.ctypes.data, out.ctypes.data, inf.ctypes.data, n) == 0
    return out, inf


def decode_points(curve, encoded, point_bytes, flags=0):
    enc = np.ascontiguousarray(encoded, dtype=np.uint8)
    n = enc.shape[0]
    out = np.zeros((n, point_bytes), dtype=np.uint8)
    ok = np.zeros((n,), dtype=np.uint8)
    inf = np.zeros((n,), dtype=np.uint8)
    assert lib().ht_decode_points(curve, enc.ctypes.data, flags, out.ctypes.data, ok.ctypes.data, inf.ctypes.data, n) == 0
    return out, ok.astype(bool), inf.astype(bool)


def fp2_sqrt(c0, c1):
    a = np.frombuffer(int(c0).to_bytes(48, "little") + int(c1).to_bytes(48, "little"), dtype=np.uint8).copy()
    r = np.zeros(96, dtype=np.uint8)
    ok = lib().ht_fp2_sqrt(a.ctypes.data, r.ctypes.data)
    b = r.tobytes()
    return bool(ok), (int.from_bytes(b[:48], "little"), int.from_bytes(b[48:], "little"))


def encode_points(curve, affine, enc_bytes):
    aff = np.ascontiguousarray(affine, dtype=np.uint8)
    n = aff.shape[0]
    out = np.zeros((n, enc_bytes), dtype=np.uint8)
    ok = np.zeros((n,), dtype=np.uint8)
    assert lib().ht_encode_points(curve, aff.ctypes.data, out.ctypes.data, ok.ctypes.data, n) == 0
    return out, ok.astype(bool)


def ntt(log2n, values, omega, flags, passes=None):
    """values: list of ints (one polynomial); returns list of ints.  passes = (t0max, tmax) shrinks the
    passes of the schedule (the device uses 10 / 8).  Asserts that the host checks of fr29.hpp saw no
    64-bit column or 32-bit limb overflow."""
    n = 1 << log2n
    a = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in values), dtype=np.uint8).copy()
    om = np.frombuffer(int(omega).to_bytes(32, "little"), dtype=np.uint8).copy()
    out = np.zeros(n * 32, dtype=np.uint8)
    if passes is None:
        assert lib().ht_ntt(log2n, om.ctypes.data, a.ctypes.data, out.ctypes.data, flags) == 0
    else:
        assert lib().ht_ntt_small_passes(log2n, om.ctypes.data, a.ctypes.data, out.ctypes.data, flags,
                                         passes[0], passes[1]) == 0
    b = out.tobytes()
    return [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(n)]


def fr29_op(op, a_limbs, b_limbs=None):
    """fr29.hpp op on raw 9-limb operands -> (9 result words, overflow count)."""
    A = np.array(list(a_limbs) + [0] * (9 - len(a_limbs)), dtype=np.uint32)
    B = np.array(b_limbs if b_limbs is not None else [0] * 9, dtype=np.uint32)
    R = np.zeros(9, dtype=np.uint32)
    ovf = lib().ht_fr29_op(op, A.ctypes.data, B.ctypes.data, R.ctypes.data)
    assert ovf >= 0
    return [int(x) for x in R], ovf


def h64_mul(a, b, portable=False):
    """bls_host64.hpp Montgomery product (R = 2^384) of two residues below p -> integer.  portable=False: the form the
    library dispatches to on this CPU (MULX / ADX when present); True: the portable form."""
    A = np.array([(a >> (64 * i)) & (2 ** 64 - 1) for i in range(6)], dtype=np.uint64)
    B = np.array([(b >> (64 * i)) & (2 ** 64 - 1) for i in range(6)], dtype=np.uint64)
    R = np.zeros(6, dtype=np.uint64)
    assert lib().ht_h64_op(2 if portable else 0, A.ctypes.data, B.ctypes.data, None, R.ctypes.data) == 0
    return sum(int(x) << (64 * i) for i, x in enumerate(R))


def h64_have_adx():
    return bool(lib().ht_h64_have_adx())


def h64_from_fe29(limbs):
    L = np.array(limbs, dtype=np.uint32)
    R = np.zeros(6, dtype=np.uint64)
    assert lib().ht_h64_op(1, None, None, L.ctypes.data, R.ctypes.data) == 0
    return sum(int(x) << (64 * i) for i, x in enumerate(R))


def ntt_plan(log2n):
    buf = np.zeros(16, dtype=np.int32)
    np_ = lib().ht_ntt_plan(log2n, buf.ctypes.data)
    return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(np_)]


def map_to_curve(curve, u, count, point_bytes):
    """u: uint8 [n, count * field bytes] -> (affine [n, point_bytes], inf [n] bool)"""
    uu = np.ascontiguousarray(u, dtype=np.uint8)
    n = uu.shape[0]
    out = np.zeros((n, point_bytes), dtype=np.uint8)
    inf = np.zeros((n,), dtype=np.uint8)
    assert lib().ht_map_to_curve(curve, uu.ctypes.data, count, out.ctypes.data, inf.ctypes.data, n) == 0
    return out, inf.astype(bool)


def fe9_op(field, op, variant, a_limbs, b_limbs):
    """Fe9 (radix-2^29 lazy field) op on raw 9-limb operands; returns the canonical integer result
    (ops 0-6, 10) or the raw word 0 (ops 7-9)."""
    A = np.array(a_limbs, dtype=np.uint32)
    B = np.array(b_limbs, dtype=np.uint32)
    R = np.zeros(8, dtype=np.uint32)
    assert lib().ht_fe9_op(field, op, variant, A.ctypes.data, B.ctypes.data, R.ctypes.data) == 0
    if op in (7, 8, 9):
        return int(R[0])
    return sum(int(R[i]) << (32 * i) for i in range(8))


def ed25519_challenge(sig, pk, msg):
    """k = SHA-512(R || A || M) mod L through the device code (csrc/sha512.hpp) on the CPU."""
    S = np.frombuffer(bytes(sig), dtype=np.uint8).copy()
    P = np.frombuffer(bytes(pk), dtype=np.uint8).copy()
    M = np.frombuffer(bytes(msg) + b"\0", dtype=np.uint8).copy()
    out = np.zeros(8, dtype=np.uint32)
    assert lib().ht_ed25519_challenge(S.ctypes.data, P.ctypes.data, M.ctypes.data, len(msg), out.ctypes.data) == 0
    return sum(int(out[i]) << (32 * i) for i in range(8))


def bls_endo_split(E, k):
    """Sub-scalars of the endomorphism MSM (csrc/endo.hpp) as signed Python ints: E = 2 (G1) or 4 (G2)."""
    a = np.array([(k >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
    o = np.zeros(E * 6, dtype=np.uint32)
    assert lib().ht_bls_endo_split(E, a.ctypes.data, o.ctypes.data) == 0
    res = []
    for e in range(E):
        v = sum(int(o[e * 6 + i]) << (32 * i) for i in range(6))
        res.append(v - (1 << 192) if v >> 191 else v)
    return res


def ecdsa_prepare(sig64, hash32, low_s=True):
    """(ok, u1, u2): the scalar side of ECDSA verification (csrc/ecdsa.hip) for one signature, on the CPU."""
    S = np.frombuffer(bytes(sig64), dtype=np.uint8).copy()
    H = np.frombuffer(bytes(hash32), dtype=np.uint8).copy()
    u1, u2 = np.zeros(8, dtype=np.uint32), np.zeros(8, dtype=np.uint32)
    ok = lib().ht_ecdsa_prepare(S.ctypes.data, H.ctypes.data, 1 if low_s else 0, u1.ctypes.data, u2.ctypes.data)
    return bool(ok), sum(int(u1[i]) << (32 * i) for i in range(8)), sum(int(u2[i]) << (32 * i) for i in range(8))


# ---- sharded MSM twin: the slot format, header check, partial-sum order and host finish of the native multi-GPU
# path (csrc/msm_shard.hpp, msm_finish.hpp, msm_plan.hpp) around a naive per-shard window-sum computation
def msm_plan(curve, n):
    out = np.zeros(4, dtype=np.int32)
    assert lib().ht_msm_plan(curve, n, out.ctypes.data) == 0
    return {"c": int(out[0]), "nwin": int(out[1]), "ngroups": int(out[2]), "acc_words": int(out[3])}


def msm_shard_slot_bytes(curve):
    return int(lib().ht_msm_shard_slot_bytes(curve))


def msm_shard_local(curve, n_local, n_max, pts_ptr, scalars_ptr):
    """pts_ptr / scalars_ptr: HOST addresses of this shard's wire arrays (the twin of the device pointers)."""
    slot = np.zeros((msm_shard_slot_bytes(curve),), dtype=np.uint8)
    assert lib().ht_msm_shard_local(curve, n_local, n_max, pts_ptr, scalars_ptr, slot.ctypes.data) == 0
    return slot


def msm_shard_windows_local(curve, n, part, nparts, pts_ptr, scalars_ptr, shared=False):
    """window-sharded mode: this part's range of the windows over ALL n points (host addresses).  shared: the slot a
    PRECOMPUTED set produces (SHARD_WINDOWS_SHARED: one grouped-sum array per rank, the ranks' slots are added)."""
    slot = np.zeros((msm_shard_slot_bytes(curve),), dtype=np.uint8)
    assert lib().ht_msm_shard_windows_local(curve, n, part, nparts, pts_ptr, scalars_ptr, slot.ctypes.data, 1 if shared else 0) == 0
    return slot


def msm_shard_combine(curve, n_max, slots, point_bytes):
    slots = np.ascontiguousarray(slots, dtype=np.uint8)
    out = np.zeros((point_bytes,), dtype=np.uint8)
    inf = np.zeros((1,), dtype=np.uint8)
    err = ctypes.create_string_buffer(400)
    rc = lib().ht_msm_shard_combine(curve, n_max, int(slots.shape[0]), slots.ctypes.data, out.ctypes.data, inf.ctypes.data, err, 400)
    if rc != 0:
        raise ValueError(err.value.decode() or "msm_shard_combine failed")
    return out, bool(inf[0])


def msm_finish(curve, c, nwin, fin_words, point_bytes, variant):
    fin = np.ascontiguousarray(fin_words, dtype=np.uint32)
    out = np.zeros((point_bytes,), dtype=np.uint8)
    inf = np.zeros((1,), dtype=np.uint8)
    assert lib().ht_msm_finish(curve, c, nwin, fin.ctypes.data, out.ctypes.data, inf.ctypes.data, variant) == 0
    return out, bool(inf[0])


def msm_plan_top(curve, n, c_override=0):
    """{c, nwin, top_tb, top_submask, vmax, nb, hprime}: the short-top-window part of the window plan (MsmPlan::top_tb)."""
    out = np.zeros(16, dtype=np.uint32)
    assert lib().ht_msm_plan_top(curve, n, c_override, out.ctypes.data) == 0
    hp = sum(int(out[6 + i]) << (32 * i) for i in range(10))
    return {"c": int(out[0]), "nwin": int(out[1]), "top_tb": int(out[2]), "top_submask": int(out[3]), "vmax": int(out[4]), "nb": int(out[5]),
            "hprime": hp}


def msm_seg(curve, n, c_override=0):
    """{c, nwin, nb, seg, nseg, lanes, accum_waves}: the lane segment msm_seg (csrc/msm_plan.hpp) picks for a whole generic plan of n points."""
    out = np.zeros(8, dtype=np.int32)
    assert lib().ht_msm_seg(curve, n, c_override, out.ctypes.data) == 0
    return dict(zip(("c", "nwin", "nb", "seg", "nseg", "lanes", "accum_waves"), [int(x) for x in out[:7]]))
