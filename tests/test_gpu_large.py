"""Beyond-BASELINE sizes on one MI355X: 2^22-point G1 MSM, 2^22 secp256k1 multiplications, 2^26-point NTT
(size-independent identities: additivity over a split of the batch, index-range invariance, round trip)."""
import time

import pytest
import torch

import bench
from noble_curves_amd import fft as G
from noble_curves_amd import _native as N
from noble_curves_amd import get_engine
from oracle import curves as OC

from helpers import wire_to_affine

pytestmark = pytest.mark.gpu


def test_large_sizes_identities():
    dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
    eng = get_engine(0)
    # 1. G1 MSM 2^22 = MSM(first half) + MSM(second half)
    m = 1 << 22
    rng = OC.makeRng(77); a, b = rng.rndBelow(OC.BLS_R - 1) + 1, rng.rndBelow(OC.BLS_R - 1) + 1
    sc = bench.gen_scalars(m, 254, 3, dev)
    pts, _ = bench.gen_points(eng, N.BLS12_381_G1, OC.BlsG1, m, a, b, dev, s)
    t0 = time.time(); r, inf = eng.msm_dev(N.BLS12_381_G1, m, pts.data_ptr(), sc.data_ptr(), s); t1 = time.time()
    h = m // 2
    r1, _ = eng.msm_dev(N.BLS12_381_G1, h, pts.data_ptr(), sc.data_ptr(), s)
    r2, _ = eng.msm_dev(N.BLS12_381_G1, h, pts[h:].data_ptr(), sc[h:].data_ptr(), s)
    P1 = OC.BlsG1.fromAffine(wire_to_affine(N.BLS12_381_G1, r1)); P2 = OC.BlsG1.fromAffine(wire_to_affine(N.BLS12_381_G1, r2))
    assert P1.add(P2).toAffine() == wire_to_affine(N.BLS12_381_G1, r), "G1 MSM 2^22 mismatch"
    print("G1 MSM 2^22 ok, %.1f ms (first call incl. workspace alloc)" % ((t1 - t0) * 1e3))
    del pts, sc
    # 2. secp mul_var 2^22 vs two halves
    m = 1 << 22
    rng = OC.makeRng(78); a, b = rng.rndBelow(OC.SECP256K1_N - 1) + 1, rng.rndBelow(OC.SECP256K1_N - 1) + 1
    sc = bench.gen_scalars(m, 255, 4, dev)
    pts, _ = bench.gen_points(eng, N.SECP256K1, OC.Secp256k1, m, a, b, dev, s)
    out = torch.empty((m, 64), dtype=torch.uint8, device=dev); inf = torch.empty((m,), dtype=torch.uint8, device=dev)
    out2 = torch.empty_like(out)
    eng.mul_var_batch_dev(N.SECP256K1, m, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), inf.data_ptr(), s)
    h = m // 2 + 12345
    eng.mul_var_batch_dev(N.SECP256K1, h, pts.data_ptr(), sc.data_ptr(), out2.data_ptr(), inf.data_ptr(), s)
    eng.mul_var_batch_dev(N.SECP256K1, m - h, pts[h:].data_ptr(), sc[h:].data_ptr(), out2[h:].data_ptr(), inf[h:].data_ptr(), s)
    torch.cuda.synchronize()
    assert bool((out == out2).all().item()), "secp 2^22 mismatch"
    print("secp256k1 multiply 2^22 ok")
    del pts, sc, out, out2
    # 3. NTT 2^26 round trip (8 GB of tables + data)
    bits = 26
    roots = G.rootsOfUnity(G.bls12_381_Fr, 7)
    x = torch.randint(0, 256, (1 << bits, 32), dtype=torch.uint8, device=dev); x[:, 31] &= 0x3F
    y = torch.empty_like(x); z = torch.empty_like(x)
    eng.ntt_dev(bits, 1, roots.omega(bits), x.data_ptr(), y.data_ptr(), s)
    eng.ntt_dev(bits, 1, roots.omega(bits), y.data_ptr(), z.data_ptr(), s, inverse=True)
    torch.cuda.synchronize()
    assert bool((z == x).all().item()), "NTT 2^26 round trip mismatch"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.ntt_dev(bits, 1, roots.omega(bits), x.data_ptr(), y.data_ptr(), s); e1.record(); torch.cuda.synchronize()
    print("NTT 2^26 ok, %.2f ms" % e0.elapsed_time(e1))
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
