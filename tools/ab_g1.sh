#!/bin/bash
# needs an A/B build of the library: make -C noble-curves_amd/csrc clean all EXTRA=-DNCG_AB_BUILD (the shipped library ignores the NCG_* variant switches, csrc/knobs.hpp)
# A/B of the bls12-381 G1 batch-multiply variants (NCG_G1_W) on the GPU box: 2^18 multiplies
for w in ${*:-141 142 151 152 132}; do
  NCG_G1_W=$w timeout 200 python - <<PY 2>/dev/null
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch, bench
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1
from oracle.curves import BLS_R, BlsG1
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0); n = 1 << 18
pts, pks = bench.gen_points(eng, BLS12_381_G1, BlsG1, n, 12345, 6789, dev, s)
sc = bench.gen_scalars(n, 254, 5, dev)
out = torch.empty((n, 96), dtype=torch.uint8, device=dev); inf = torch.empty((n,), dtype=torch.uint8, device=dev)
f = lambda: eng.mul_var_batch_dev(BLS12_381_G1, n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), inf.data_ptr(), s)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): f()
e1.record(); torch.cuda.synchronize()
ks = bench.scalars_to_ints(sc)
ones = torch.zeros((n, 32), dtype=torch.uint8, device=dev); ones[:, 0] = 1
tot, _ = eng.msm_dev(BLS12_381_G1, n, out.data_ptr(), ones.data_ptr(), s)
from helpers import wire_to_affine
ok = wire_to_affine(BLS12_381_G1, tot) == BlsG1.BASE.multiplyUnsafe(sum(k * p for k, p in zip(ks, pks)) % BLS_R).toAffine()
print("NCG_G1_W=$w", round(e0.elapsed_time(e1) / 3, 3), "ms", "checksum ok" if ok else "CHECKSUM MISMATCH")
PY
done
