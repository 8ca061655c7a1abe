#!/usr/bin/env python3
"""A few window-sharded local phases (part 0 of G) for a kernel trace: 
    rocprofv3 --kernel-trace --output-format csv -d out -- python tools/share_once.py --parts 8 --kind resident
then tools/msm_timeline.py out --seq."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, bench
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
from oracle.curves import BlsG1, BlsG2
ap = argparse.ArgumentParser()
ap.add_argument("--curve", default="g1"); ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--parts", type=int, default=8); ap.add_argument("--kind", default="resident"); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--part", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
cid, O = (BLS12_381_G1, BlsG1) if a.curve == "g1" else (BLS12_381_G2, BlsG2)
n = 1 << a.log2n
pts, ks = bench.gen_points(eng, cid, O, n, 0x1234567 + 7, 0x6789, dev, s)
sc = bench.gen_scalars(n, 254, 5, dev); sc[::17] = 0
rs = None
if a.kind != "generic":
    rs = eng.upload_points(cid, pts.cpu().numpy())
    if a.kind.startswith("verified"): assert rs.verify_subgroup() == -1
    if a.kind.endswith("precomputed") or a.kind.endswith("precomp"): assert rs.precompute()
P = 0 if rs is not None else pts.data_ptr()
for _ in range(a.reps):
    if a.parts <= 1:
        (rs.msm_dev(sc.data_ptr(), s) if rs is not None else eng.msm_dev(cid, n, P, sc.data_ptr(), s))
    else:
        eng.msm_shard_windows_local_dev(cid, n, a.part, a.parts, P, sc.data_ptr(), s, rs)
    torch.cuda.synchronize()
print("plan", eng.msm_last_plan())
