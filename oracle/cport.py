"""ctypes loader of the oracle's C restatement (oracle/c/oracle.c -> oracle/_build/liboracle.so).
TEST INFRASTRUCTURE: used by tests/ and bench.py's cpu_baseline leg only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            subprocess.check_call(["make", "-C", os.path.join(_HERE, "c")], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(SO)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        for name in ("orc_secp256k1_multiply_unsafe", "orc_bls12_381_g1_multiply_unsafe"):
            getattr(L, name).argtypes = [vp, vp, vp, vp, sz]
        for name in ("orc_bls12_381_g1_pippenger", "orc_secp256k1_pippenger", "orc_bls12_381_g2_pippenger"):
            getattr(L, name).argtypes = [vp, vp, sz, vp, vp]
        L.orc_secp256k1_multiply.argtypes = [vp, vp, vp, vp, vp, sz]
        L.orc_ed25519_verify_batch.argtypes = [vp, vp, vp, ctypes.c_int, vp, sz]
        L.orc_fft_fr.argtypes = [ctypes.c_int, vp, vp, vp, ctypes.c_int]
        _lib = L
    return _lib


def _u8(a, width):
    return np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, width)


def multiply_unsafe(curve_name, pts_wire, scalars_wire):
    pb = {"secp256k1": 64, "bls12_381_g1": 96}[curve_name]
    pts, sc = _u8(pts_wire, pb), _u8(scalars_wire, 32)
    n = pts.shape[0]
    out = np.zeros((n, pb), np.uint8)
    inf = np.zeros((n,), np.uint8)
    fn = getattr(lib(), "orc_%s_multiply_unsafe" % curve_name)
    fn(pts.ctypes.data, sc.ctypes.data, out.ctypes.data, inf.ctypes.data, n)
    return out, inf


def multiply(pts_wire, scalars_wire, blinds=None):
    """secp256k1 Point.multiply (constant-time fixed-window shape of the reference; blinds: uint8 [n, 16] = the
    reference's randomBytes(16) per call, or None for the unblinded shape)."""
    pts, sc = _u8(pts_wire, 64), _u8(scalars_wire, 32)
    n = pts.shape[0]
    out = np.zeros((n, 64), np.uint8)
    inf = np.zeros((n,), np.uint8)
    bl = None if blinds is None else _u8(blinds, 16)
    lib().orc_secp256k1_multiply(pts.ctypes.data, sc.ctypes.data, None if bl is None else bl.ctypes.data, out.ctypes.data,
                                 inf.ctypes.data, n)
    return out, inf


def pippenger(curve_name, pts_wire, scalars_wire):
    pb = {"secp256k1": 64, "bls12_381_g1": 96, "bls12_381_g2": 192}[curve_name]
    pts, sc = _u8(pts_wire, pb), _u8(scalars_wire, 32)
    n = pts.shape[0]
    out = np.zeros((pb,), np.uint8)
    inf = np.zeros((1,), np.uint8)
    fn = getattr(lib(), "orc_%s_pippenger" % curve_name)
    fn(pts.ctypes.data, sc.ctypes.data, n, out.ctypes.data, inf.ctypes.data)
    return out, bool(inf[0])


def ed25519_verify_batch(sigs, pks, ks, zip215=True):
    sigs, pks, ks = _u8(sigs, 64), _u8(pks, 32), _u8(ks, 32)
    n = sigs.shape[0]
    out = np.zeros((n,), np.uint8)
    lib().orc_ed25519_verify_batch(sigs.ctypes.data, pks.ctypes.data, ks.ctypes.data, 1 if zip215 else 0,
                                   out.ctypes.data, n)
    return out.astype(bool)


def fft_fr(bits, data, omega, inverse=False, brp_input=False, brp_output=False):
    """FFT(roots, Fr).direct / .inverse on uint8 [2^bits, 32] canonical LE residues (fft.ts:518-577)."""
    d = _u8(data, 32)
    assert d.shape[0] == 1 << bits
    out = np.empty_like(d)
    om = np.frombuffer(int(omega).to_bytes(32, "little"), dtype=np.uint8).copy()
    flags = (1 if inverse else 0) | (2 if brp_input else 0) | (4 if brp_output else 0)
    rc = lib().orc_fft_fr(bits, om.ctypes.data, d.ctypes.data, out.ctypes.data, flags)
    assert rc == 0
    return out
