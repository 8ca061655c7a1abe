import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, bench, time
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
from oracle.curves import BlsG1, BlsG2
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for cid, O, lg in ((BLS12_381_G1, BlsG1, 20), (BLS12_381_G2, BlsG2, 18)):
    if which == "g1" and cid != BLS12_381_G1: continue
    if which == "g2" and cid != BLS12_381_G2: continue
    n = 1 << lg
    pts, _ = bench.gen_points(eng, cid, O, n, 12345, 6789, dev, s)
    sc = bench.gen_scalars(n, 254, 5, dev)
    res = eng.upload_points(cid, pts.cpu().numpy())
    assert res.verify_subgroup() == -1
    f = lambda: res.msm_dev(sc.data_ptr(), s)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print("curve", cid, "endo msm", round(e0.elapsed_time(e1) / 5, 3), "ms", flush=True)
