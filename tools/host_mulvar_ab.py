#!/usr/bin/env python3
"""Host-pointer secp256k1 batch multiply (pinned buffers, 2^20 pairs): chunk pattern 1 : 3 : 3 : 1 against four equal chunks - needs an
A/B build (the shipped library ignores NCG_MULVAR_HOST_EVEN).   NCG_LIB=$PWD/tools/_build/libncg_ab.so python tools/host_mulvar_ab.py"""
import os, sys, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch, bench
    from noble_curves_amd import get_engine
    from noble_curves_amd._native import SECP256K1
    from oracle.curves import Secp256k1
    dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
    eng = get_engine(0)
    n = 1 << 20
    pts, _ = bench.gen_points(eng, SECP256K1, Secp256k1, n, 0x1234567, 0x6789, dev, st.cuda_stream)
    sc = bench.gen_scalars(n, 255, 5, dev)
    hp, hs = pts.cpu().numpy(), sc.cpu().numpy()
    out = np.zeros((n, 64), np.uint8); inf = np.zeros((n,), np.uint8)
    for a in (hp, hs, out, inf): eng.host_register(a)
    call = lambda: eng._check(eng.lib.ncg_mul_var_batch(eng.h, SECP256K1, n, hp.ctypes.data, hs.ctypes.data, out.ctypes.data, inf.ctypes.data))
    for _ in range(3): call()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(json.dumps({"ms": round(ts[len(ts) // 2], 3), "min": round(ts[0], 3)}))
    sys.exit(0)
for rep in range(2):
    for even in (1, 0):
        env = dict(os.environ, NCG_MULVAR_HOST_EVEN=str(even))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        print("even" if even else "1:3:3:1", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
