import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, bench, time
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
from oracle.curves import BlsG1, BlsG2
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
for cid, O, lg in ((BLS12_381_G1, BlsG1, 20), (BLS12_381_G2, BlsG2, 18)):
    n = 1 << lg
    pts, _ = bench.gen_points(eng, cid, O, n, 12345, 6789, dev, s)
    sc = bench.gen_scalars(n, 254, 5, dev)
    f = lambda: eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr(), s)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    print("curve", cid, "msm", round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms", flush=True)
