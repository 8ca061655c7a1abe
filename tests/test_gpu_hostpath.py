"""Host-pointer entry points at sizes where they run in chunks / parts (include/ncg.h conventions: host buffers in, host
buffers out): ncg_msm uploads the scalars first and the points in PARTS, each part accumulated into the shared buckets
while the next is on the bus (csrc/api.hip, MsmPlan::part_flags); ncg_mul_var_batch pipelines H2D / kernels / D2H over
chunks.  Results must equal the device-buffer entry points bit for bit (pippenger, curve.ts:863-905, is a sum over
points: the parts add up), with pageable and with pinned-once buffers, skewed scalars, ZERO members, and a scalar
outside the group order in a later part (validateMSMScalars, curve.ts:398-404: the global index is reported)."""
import numpy as np
import pytest
import torch

import bench
from helpers import ORACLE_CURVE, wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, ED25519, POINT_BYTES, SECP256K1

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve,n", [(BLS12_381_G1, (1 << 17) + 1000), (BLS12_381_G1, (1 << 19) + 5), (SECP256K1, (1 << 17) + 3),
                                     (ED25519, 1 << 17), (BLS12_381_G2, (1 << 17) + 9), (BLS12_381_G1, (1 << 16) + 1)])
def test_host_pointer_msm_in_parts_equals_device_msm(curve, n):
    eng = get_engine()
    dev = torch.device("cuda", 0)
    Pt = ORACLE_CURVE[curve]
    order = Pt.Fn.ORDER
    if curve == ED25519:
        pts, ks = bench.gen_points(eng, curve, Pt, n, 0x51ED, 0x77, dev, None)
    else:
        pts, ks = bench.gen_points(eng, curve, Pt, n, 0xABCDEF1 + curve, 0x1357, dev, None)
    pts_h = pts.cpu().numpy().copy()
    pts_h[n // 2 + 3] = 0                              # ZERO member (Weierstrass wire form) in a later part
    if curve == ED25519:
        pts_h[n // 2 + 3, 32] = 1                      # Edwards identity (0, 1)
    cases = []
    sc = bench.gen_scalars(n, 250, 11 + curve, dev).cpu().numpy().copy()
    sc[::17] = 0
    cases.append(sc)
    same = np.tile(np.frombuffer(int(0xDEADBEEFCAFEF00D1234567 % order).to_bytes(32, "little"), dtype=np.uint8), (n, 1)).copy()
    cases.append(same)                                 # one bucket per window holds every entry of every part
    d_pts = torch.from_numpy(pts_h).to(dev)
    for k, sc_h in enumerate(cases):
        want, want_inf = eng.msm_dev(curve, n, d_pts.data_ptr(), torch.from_numpy(sc_h).to(dev).data_ptr())
        got, inf = eng.msm(curve, pts_h, sc_h)                 # pageable buffers: registered for the call
        assert np.array_equal(got, want) and inf == want_inf, (k, "pageable")
        eng.host_register(pts_h)
        eng.host_register(sc_h)
        try:
            got, inf = eng.msm(curve, pts_h, sc_h)             # pinned once
            assert np.array_equal(got, want) and inf == want_inf, (k, "pinned")
        finally:
            eng.host_unregister(pts_h)
            eng.host_unregister(sc_h)
    # the progression identity pins the VALUE itself on every curve (not only product == product): P_i = k_i G, so the MSM is
    # (sum k_i s_i mod order) G, computed by the oracle (test/slow-curves.test.ts:185-252 builds its expected value this way)
    for sc_h in cases:
        sci = [int.from_bytes(sc_h[i].tobytes(), "little") for i in range(n)]
        sci[n // 2 + 3] = 0                                    # the ZERO member contributes nothing
        exp = Pt.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sci)) % order).toAffine()
        assert wire_to_affine(curve, eng.msm(curve, pts_h, sc_h)[0]) == exp
    bad = cases[0].copy()
    idx = n - 5                                                # in the last part
    bad[idx] = np.frombuffer(int(order).to_bytes(32, "little"), dtype=np.uint8)
    with pytest.raises(Exception, match="invalid scalar at index %d" % idx):
        eng.msm(curve, pts_h, bad)
    got, _ = eng.msm(curve, pts_h, cases[0])                   # the context keeps working
    assert np.array_equal(got, eng.msm_dev(curve, n, d_pts.data_ptr(), torch.from_numpy(cases[0]).to(dev).data_ptr())[0])


@pytest.mark.parametrize("curve,n", [(SECP256K1, (1 << 17) + 77), (BLS12_381_G1, 1 << 17), (SECP256K1, (1 << 19) + 1)])
def test_host_pointer_batch_multiply_in_chunks_equals_device_batch(curve, n):
    eng = get_engine()
    dev = torch.device("cuda", 0)
    Pt = ORACLE_CURVE[curve]
    pts, ks = bench.gen_points(eng, curve, Pt, n, 0x77AA + curve, 0x9, dev, None)
    sc = bench.gen_scalars(n, 250, 5, dev, edge_order=Pt.Fn.ORDER)
    out = torch.empty_like(pts)
    inf = torch.empty((n,), dtype=torch.uint8, device=dev)
    eng.mul_var_batch_dev(curve, n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), inf.data_ptr())
    torch.cuda.synchronize()
    o, i = eng.mul_var_batch(curve, pts.cpu().numpy(), sc.cpu().numpy())
    assert np.array_equal(o, out.cpu().numpy()) and np.array_equal(i, inf.cpu().numpy())
    # ... and the values themselves against the oracle on a sample that touches every chunk: s_i P_i = (s_i k_i mod order) G
    sc_h = sc.cpu().numpy()
    for j in list(range(0, n, n // 13)) + [n - 1]:
        sj = int.from_bytes(sc_h[j].tobytes(), "little")
        want = Pt.BASE.multiplyUnsafe(sj * ks[j] % Pt.Fn.ORDER)
        assert wire_to_affine(curve, o[j]) == want.toAffine() and bool(i[j]) == want.is0(), j
    # the same call into result arrays the caller keeps and pins once (the chunk pattern 1 : 3 : 3 : 1 at n >= 2^19)
    ko, ki = np.zeros((n, POINT_BYTES[curve]), np.uint8), np.zeros((n,), np.uint8)
    eng.host_register(ko)
    try:
        o2, i2 = eng.mul_var_batch(curve, pts.cpu().numpy(), sc.cpu().numpy(), out=ko, inf=ki)
    finally:
        eng.host_unregister(ko)
    assert o2 is ko and i2 is ki and np.array_equal(ko, o) and np.array_equal(ki, i)
    with pytest.raises(ValueError, match="out must be"):
        eng.mul_var_batch(curve, pts.cpu().numpy(), sc.cpu().numpy(), out=ko[:-1])
