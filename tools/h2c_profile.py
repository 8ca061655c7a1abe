#!/usr/bin/env python3
"""bls12-381 hash-to-curve map (device part) alone, a few launches of G1 (2^18) and G2 (2^16) with count = 2:
the target of `rocprofv3 --kernel-trace --stats` when the per-kernel split of the map is wanted."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from noble_curves_amd import get_engine  # noqa: E402
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    eng = get_engine(0)
    P = lambda t: t.data_ptr()  # noqa: E731
    for curve, n, words in ((BLS12_381_G1, 1 << 18, 2), (BLS12_381_G2, 1 << 16, 4)):
        u = torch.randint(0, 256, (n, words * 48), dtype=torch.uint8, device=dev)
        for k in range(words):
            u[:, 48 * k + 47] &= 0x0F
        out = torch.empty((n, 96 if curve == BLS12_381_G1 else 192), dtype=torch.uint8, device=dev)
        inf = torch.empty((n,), dtype=torch.uint8, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng.map_to_curve_batch_dev(curve, n, 2, P(u), P(out), P(inf), s)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            eng.map_to_curve_batch_dev(curve, n, 2, P(u), P(out), P(inf), s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("curve %d n=2^%d count=2: %.3f ms, %.3e points/s" % (curve, n.bit_length() - 1, ms, n / ms * 1e3), flush=True)


if __name__ == "__main__":
    main()
