#!/usr/bin/env python3
"""End-to-end rate of the host-pointer entry points (inputs and outputs in host memory, pageable numpy
buffers): H2D + kernels + D2H per call, next to the device-resident numbers of bench.py."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from noble_curves_amd import get_engine  # noqa: E402
from noble_curves_amd._native import BLS12_381_G1, SECP256K1  # noqa: E402
from oracle.curves import BLS_R, BlsG1, SECP256K1_N, Secp256k1, makeRng  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    eng = get_engine(0)
    n = 1 << 20
    res = {}
    rng = makeRng(5)
    a, b = rng.rndBelow(SECP256K1_N - 1) + 1, rng.rndBelow(SECP256K1_N - 1) + 1
    pts, _ = bench.gen_points(eng, SECP256K1, Secp256k1, n, a, b, dev, st.cuda_stream)
    sc = bench.gen_scalars(n, 255, 5, dev)
    hp, hs = pts.cpu().numpy(), sc.cpu().numpy()
    eng.mul_var_batch(SECP256K1, hp, hs)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.mul_var_batch(SECP256K1, hp, hs)
    dt = (time.perf_counter() - t0) / 3
    res["secp256k1 multiplyUnsafe batch, host buffers (fresh output array per call)"] = {"n": n, "ms": dt * 1e3, "per_s": n / dt}
    out = np.zeros((n, 64), np.uint8)                      # reused, already touched output buffers
    inf = np.zeros((n,), np.uint8)
    call = lambda: eng._check(eng.lib.ncg_mul_var_batch(eng.h, SECP256K1, n, hp.ctypes.data, hs.ctypes.data,  # noqa: E731
                                                        out.ctypes.data, inf.ctypes.data))
    call()
    t0 = time.perf_counter()
    for _ in range(3):
        call()
    dt = (time.perf_counter() - t0) / 3
    res["secp256k1 multiplyUnsafe batch, host buffers"] = {"n": n, "ms": dt * 1e3, "per_s": n / dt}
    a, b = rng.rndBelow(BLS_R - 1) + 1, rng.rndBelow(BLS_R - 1) + 1
    g1, _ = bench.gen_points(eng, BLS12_381_G1, BlsG1, n, a, b, dev, st.cuda_stream)
    gs = bench.gen_scalars(n, 254, 6, dev)
    hg, hgs = g1.cpu().numpy(), gs.cpu().numpy()
    eng.msm(BLS12_381_G1, hg, hgs)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.msm(BLS12_381_G1, hg, hgs)
    dt = (time.perf_counter() - t0) / 3
    res["bls12-381 G1 MSM, host buffers"] = {"n": n, "ms": dt * 1e3, "per_s": n / dt}
    # the same calls on buffers pinned ONCE (ncg_host_register): what a binding does for the buffers it reuses
    for arr in (hp, hs, out, inf, hg, hgs):
        eng.host_register(arr)
    call()
    t0 = time.perf_counter()
    for _ in range(5):
        call()
    dt = (time.perf_counter() - t0) / 5
    res["secp256k1 multiplyUnsafe batch, host buffers pinned once"] = {"n": n, "ms": dt * 1e3, "per_s": n / dt}
    o1, _ = eng.msm(BLS12_381_G1, hg, hgs)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.msm(BLS12_381_G1, hg, hgs)
    dt = (time.perf_counter() - t0) / 5
    res["bls12-381 G1 MSM, host buffers pinned once"] = {"n": n, "ms": dt * 1e3, "per_s": n / dt}
    o2, _ = eng.msm_dev(BLS12_381_G1, n, g1.data_ptr(), gs.data_ptr())
    assert (o1 == o2).all(), "host-buffer MSM differs from the device-buffer MSM"
    ob = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    ib = torch.empty((n,), dtype=torch.uint8, device=dev)
    eng.mul_var_batch_dev(SECP256K1, n, pts.data_ptr(), sc.data_ptr(), ob.data_ptr(), ib.data_ptr())
    torch.cuda.synchronize()
    assert (ob.cpu().numpy() == out).all() and (ib.cpu().numpy() == inf).all(), "host-buffer batch multiply differs from the device-buffer one"
    # wire time alone: the same bytes by plain copies between the pinned arrays and device memory
    dbuf = torch.empty((hg.nbytes + hgs.nbytes,), dtype=torch.uint8, device=dev)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    t0 = time.perf_counter()
    for _ in range(5):
        hip.hipMemcpy(dbuf.data_ptr(), hg.ctypes.data, hg.nbytes, 1)
        hip.hipMemcpy(dbuf.data_ptr() + hg.nbytes, hgs.ctypes.data, hgs.nbytes, 1)
    dt = (time.perf_counter() - t0) / 5
    res["H2D of the MSM's 128 MB from pinned memory (wire time)"] = {"ms": dt * 1e3, "GB_per_s": (hg.nbytes + hgs.nbytes) / dt / 1e9}
    for arr in (hp, hs, out, inf, hg, hgs):
        eng.host_unregister(arr)
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
