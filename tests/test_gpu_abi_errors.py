"""C-ABI error conventions on a live context (status codes, never exceptions; messages through
ncg_last_error): NULL buffers, unknown curves, out-of-range sizes, bad roots - and that a failed call leaves
the context usable."""
import ctypes

import numpy as np
import pytest

from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, ED25519, SECP256K1, NativeError

pytestmark = pytest.mark.gpu
OK, INVALID, UNSUPPORTED = 0, -1, -4


def _err(eng):
    return (eng.lib.ncg_last_error(eng.h) or b"").decode()


def test_status_codes_and_messages():
    eng = get_engine()
    L, h = eng.lib, eng.h
    buf = np.zeros(4096, dtype=np.uint8)
    p = buf.ctypes.data
    # unknown curve ids
    assert L.ncg_mul_var_batch(h, 9, 1, p, p, p, p) == UNSUPPORTED and "unsupported curve 9" in _err(eng)
    assert L.ncg_msm(h, -1, 1, p, p, p, None) == UNSUPPORTED
    assert L.ncg_decode_points_batch(h, 7, 1, p, 0, p, p, p) == UNSUPPORTED
    assert L.ncg_encode_points_batch(h, 7, 1, p, p, p) == UNSUPPORTED
    assert L.ncg_map_to_curve_batch(h, SECP256K1, 1, 1, p, p, p) == UNSUPPORTED
    assert L.ncg_ntt(h, 3, 2, 1, p, p, p, 0) == UNSUPPORTED
    # NULL buffers
    assert L.ncg_mul_var_batch(h, SECP256K1, 1, None, p, p, p) == INVALID and "NULL buffer" in _err(eng)
    assert L.ncg_mul_base_batch(h, SECP256K1, 1, None, p, p) == INVALID
    assert L.ncg_msm(h, BLS12_381_G1, 2, p, None, p, None) == INVALID
    assert L.ncg_ed25519_verify_batch(h, 1, p, p, None, 1, p) == INVALID
    assert L.ncg_ntt(h, 0, 2, 1, None, p, p, 0) == INVALID
    # out-of-range parameters
    assert L.ncg_ntt(h, 0, 29, 1, p, p, p, 0) == INVALID and "log2n 29 out of range" in _err(eng)
    assert L.ncg_ntt(h, 0, -1, 1, p, p, p, 0) == INVALID
    assert L.ncg_map_to_curve_batch(h, BLS12_381_G1, 1, 3, p, p, p) == INVALID and "count must be 1 or 2" in _err(eng)
    # empty batches are successes and touch nothing
    assert L.ncg_mul_var_batch(h, SECP256K1, 0, None, None, None, None) == OK
    assert L.ncg_decode_points_batch(h, ED25519, 0, None, 0, None, None, None) == OK
    assert L.ncg_ntt(h, 0, 3, 0, p, None, None, 0) == OK
    bad = ctypes.c_int64(7)
    out = np.zeros(96, dtype=np.uint8)
    inf = ctypes.c_uint8(9)
    assert L.ncg_aggregate_encoded(h, BLS12_381_G1, 0, None, 0, out.ctypes.data, ctypes.byref(inf), ctypes.byref(bad)) == OK
    assert bad.value == -1 and inf.value == 1
    # a NULL context never dereferences
    assert L.ncg_mul_var_batch(None, SECP256K1, 1, p, p, p, p) == INVALID
    # a wrong root of unity is refused, and the context keeps working afterwards
    with pytest.raises(NativeError, match="primitive 2\\^3-th root"):
        eng.ntt(3, np.zeros((8, 32), np.uint8), 12345)
    k = np.zeros((1, 32), dtype=np.uint8)
    k[0, 0] = 1
    o, f = eng.mul_base_batch(SECP256K1, k)
    assert int.from_bytes(o[0, :32].tobytes(), "little") == 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798 and not f[0]


def test_msm_rejects_scalars_not_below_the_group_order():
    """validateMSMScalars (curve.ts:398-404): the window plan covers scalars below the order only, so the
    C ABI refuses anything else instead of returning a wrong sum (the shims reject earlier, with the
    reference's message)."""
    import numpy as np
    from helpers import points_to_wire, scalars_to_wire
    from noble_curves_amd import get_engine
    from noble_curves_amd._native import BLS12_381_G1, NativeError
    from oracle.curves import BLS_R, BlsG1
    eng = get_engine()
    pts = points_to_wire(BLS12_381_G1, [BlsG1.BASE.multiplyUnsafe(i + 1) for i in range(40)])
    ok = scalars_to_wire([(i * 7919 + 3) % BLS_R for i in range(40)])
    eng.msm(BLS12_381_G1, pts, ok)
    for bad_k in (BLS_R, BLS_R + 5, (1 << 256) - 1):
        sc = ok.copy()
        sc[17] = np.frombuffer(int(bad_k).to_bytes(32, "little"), np.uint8)
        with pytest.raises(NativeError, match="invalid scalar at index 17"):
            eng.msm(BLS12_381_G1, pts, sc)
    eng.msm(BLS12_381_G1, pts, ok)          # the context survives
