#!/usr/bin/env python3
"""Time of the secp256k1 batch multiply against the batch size around 2^20, to see the occupancy quantisation: at three waves
per SIMD the chip holds 768 workgroups of 256 lanes = 196 608 items at a time, so 2^20 items are 5.33 "rounds".
    python tools/secp_rounds.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from noble_curves_amd import get_engine
from noble_curves_amd._native import SECP256K1
from oracle.curves import Secp256k1, SECP256K1_N
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
NMAX = 1572864
pts, _ = bench.gen_points(eng, SECP256K1, Secp256k1, NMAX, 0x1234567, 0x6789, dev, s)
sc = bench.gen_scalars(NMAX, 255, 99, dev, edge_order=SECP256K1_N)
out = torch.empty((NMAX, 64), dtype=torch.uint8, device=dev); inf = torch.empty((NMAX,), dtype=torch.uint8, device=dev)
res = []
for n in [196608 * k for k in (1, 2, 3, 4, 5, 6, 7, 8)] + [1 << 20, 131072 * 8, 917504, 1114112]:
    for _ in range(3):
        eng.mul_var_batch_dev(SECP256K1, n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), inf.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 8
    for _ in range(K):
        eng.mul_var_batch_dev(SECP256K1, n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), inf.data_ptr(), s)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    res.append({"n": n, "rounds_at_3_waves": round(n / 196608, 3), "ms": round(ms, 3), "ns_per_item": round(ms * 1e6 / n, 2)})
    print(res[-1])
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=0)
