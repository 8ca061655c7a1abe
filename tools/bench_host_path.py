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
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
