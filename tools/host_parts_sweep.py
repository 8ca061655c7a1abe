#!/usr/bin/env python3
"""Host-pointer G1 / G2 MSM (pinned buffers) against the number of parts the points and scalars cross the bus in - needs an A/B
build of the library (NCG_LIB=tools/_build/libncg_ab.so; the shipped one ignores NCG_MSM_HOST_PARTS).
    NCG_LIB=$PWD/tools/_build/libncg_ab.so python tools/host_parts_sweep.py"""
import os, sys, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch, bench
    from noble_curves_amd import get_engine
    from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
    from oracle.curves import BlsG1, BlsG2
    dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
    eng = get_engine(0)
    out = {}
    for name, cid, O, n in (("g1", BLS12_381_G1, BlsG1, 1 << 20), ("g2", BLS12_381_G2, BlsG2, 1 << 18)):
        pts, _ = bench.gen_points(eng, cid, O, n, 0x1234567, 0x6789, dev, st.cuda_stream)
        sc = bench.gen_scalars(n, 254, 5, dev)
        hp, hs = pts.cpu().numpy(), sc.cpu().numpy()
        eng.host_register(hp); eng.host_register(hs)
        for _ in range(3): eng.msm(cid, hp, hs)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); eng.msm(cid, hp, hs); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        out[name] = round(ts[len(ts) // 2], 3)
        eng.host_unregister(hp); eng.host_unregister(hs)
    print(json.dumps(out))
    sys.exit(0)
for rep in range(2):
    for parts in (1, 2, 4, 8):
        env = dict(os.environ, NCG_MSM_HOST_PARTS=str(parts))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        print("parts", parts, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
