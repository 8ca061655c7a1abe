#!/usr/bin/env python3
"""NTT over bls12-381 Fr: time per transform for several sizes / orderings with device-resident
data, HIP-event timing on an explicit stream.  Reports elements/s and the effective HBM rate
N * 64 B / t (one read + one write of the data is the algorithmic traffic).  Every size is checked
by inverse(direct(x)) == x on the device before it is timed."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from noble_curves_amd import fft as G  # noqa: E402
from noble_curves_amd import get_engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--sizes", default="12,16,20,22,24")
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    eng = get_engine(0)
    roots = G.rootsOfUnity(G.bls12_381_Fr, 7)
    res = {}
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    for bits in [int(x) for x in args.sizes.split(",")]:
        n = 1 << bits
        batch = max(1, (1 << 22) >> bits) if bits < 22 else 1
        x = torch.randint(0, 256, (batch * n, 32), dtype=torch.uint8, device=dev, generator=gen)
        x[:, 31] &= 0x3F                                   # < 2^254 < r: canonical residues
        y = torch.empty_like(x)
        z = torch.empty_like(x)
        om = roots.omega(bits)
        eng.ntt_dev(bits, batch, om, x.data_ptr(), y.data_ptr(), s)
        eng.ntt_dev(bits, batch, om, y.data_ptr(), z.data_ptr(), s, inverse=True)
        torch.cuda.synchronize()
        assert bool((z == x).all().item()), "inverse(direct(x)) != x at 2^%d" % bits
        eng.ntt_dev(bits, batch, om, x.data_ptr(), y.data_ptr(), s, brp_output=True)
        eng.ntt_dev(bits, batch, om, y.data_ptr(), z.data_ptr(), s, inverse=True, brp_input=True)
        torch.cuda.synchronize()
        assert bool((z == x).all().item()), "brp round trip failed at 2^%d" % bits
        for name, kw in (("natural->natural", {}), ("natural->bitrev", {"brp_output": True}),
                         ("bitrev->natural inverse", {"inverse": True, "brp_input": True})):
            fn = lambda: eng.ntt_dev(bits, batch, om, x.data_ptr(), y.data_ptr(), s, **kw)  # noqa: E731
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.steps
            key = "2^%d x%d %s" % (bits, batch, name)
            res[key] = {"log2n": bits, "batch": batch, "ms": round(ms, 4), "elements_per_s": batch * n / (ms * 1e-3),
                        "butterflies_per_s": batch * n / 2 * bits / (ms * 1e-3),
                        "algorithmic_GBps": batch * n * 64 / (ms * 1e-3) / 1e9}
            print("%-40s %9.4f ms  %.3e elem/s  %.3e bfly/s  %7.1f GB/s (N*64B/t)" % (
                key, ms, res[key]["elements_per_s"], res[key]["butterflies_per_s"], res[key]["algorithmic_GBps"]), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
