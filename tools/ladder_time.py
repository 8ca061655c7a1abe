"""Time of the secp256k1 batch multiply on 2^20 pseudo-random (x, y) pairs WITHOUT any verification: for timing-experiment builds whose
results are wrong on purpose (NCG_LIB=...; the formulas never test curve membership, so the instruction stream is the shipped one).
    python tools/ladder_time.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from noble_curves_amd import get_engine
from noble_curves_amd._native import SECP256K1
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
n = 1 << 20
g = torch.Generator(device="cpu"); g.manual_seed(11)
pts = torch.randint(0, 256, (n, 64), dtype=torch.uint8, generator=g)
pts[:, 31] &= 0x7F; pts[:, 63] &= 0x7F                      # x, y below 2^255 < p
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g); sc[:, 31] &= 0x7F
pts, sc = pts.to(dev), sc.to(dev)
out = torch.empty((n, 64), dtype=torch.uint8, device=dev); inf = torch.empty((n,), dtype=torch.uint8, device=dev)
for _ in range(10):
    eng.mul_var_batch_dev(SECP256K1, n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), inf.data_ptr(), s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.mul_var_batch_dev(SECP256K1, n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), inf.data_ptr(), s)
torch.cuda.synchronize()
print("secp256k1 2^20 batch multiply: %.3f ms per step over %d steps (%s)" % ((time.perf_counter() - t0) / steps * 1e3, steps, os.environ.get("NCG_LIB", "shipped library")))
