#!/usr/bin/env python3
"""A/B of the MSM tail components on the GPU (one process; the NCG_MSM_* switches are read per call): which sizes
still satisfy the progression identity (test/slow-curves.test.ts:185-252) under which switch, with wall times.
Needs an A/B build (make -C noble-curves_amd/csrc clean all EXTRA=-DNCG_AB_BUILD): the shipped library ignores these switches."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, bench
from helpers import wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, SECP256K1
from oracle.curves import BLS_R, BlsG1, BlsG2, SECP256K1_N, Secp256k1
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
CONFIGS = [{}, {"NCG_MSM_COOP": "1", "NCG_MSM_COOP_LEVEL": "0"}, {"NCG_MSM_COOP": "0"},
           {"NCG_MSM_COOP": "1", "NCG_MSM_COOP_LEVEL": "1", "NCG_MSM_RUN_SERIAL": "1000000"},
           {"NCG_MSM_COOP": "0", "NCG_MSM_RUN_SERIAL": "1000000"}]
KEYS = ["NCG_MSM_COOP", "NCG_MSM_COOP_LEVEL", "NCG_MSM_RUN_SERIAL"]
curves = sys.argv[1].split(",") if len(sys.argv) > 1 else ["g1", "g2", "secp"]
for name, cid, O, order, top in (("g1", BLS12_381_G1, BlsG1, BLS_R, 17), ("g2", BLS12_381_G2, BlsG2, BLS_R, 14), ("secp", SECP256K1, Secp256k1, SECP256K1_N, 14)):
    if name not in curves:
        continue
    n = 1 << top
    pts, ks = bench.gen_points(eng, cid, O, n, 0x1234567, 0x6789, dev, s)
    sc = bench.gen_scalars(n, 250, 5, dev)
    sc[::17] = 0
    sci = bench.scalars_to_ints(sc)
    print("generated", name, flush=True)
    for lg in range(6, top + 1):
        m = 1 << lg
        expect = O.BASE.multiplyUnsafe(sum(k * x for k, x in zip(ks[:m], sci[:m])) % order).toAffine()
        for cfg in CONFIGS:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(cfg)
            t0 = time.perf_counter()
            r = eng.msm_dev(cid, m, pts.data_ptr(), sc.data_ptr(), s)
            dt = (time.perf_counter() - t0) * 1e3
            ok = wire_to_affine(cid, r[0]) == expect
            print("%s 2^%d %s %s %.2f ms" % (name, lg, "ok  " if ok else "FAIL", json.dumps(cfg), dt), flush=True)
