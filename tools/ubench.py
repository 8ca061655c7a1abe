#!/usr/bin/env python3
"""Instruction-rate micro-benchmarks on the GPU (feeds DESIGN.md's VALU roofline)."""
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from noble_curves_amd import get_engine  # noqa: E402

KINDS = {0: ("v_mad_u64_u32", 8), 1: ("v_mul_lo_u32", 8), 2: ("v_mul_hi_u32(+add)", 8), 3: ("v_mad_u32_u24", 8),
         12: ("v_mul_hi_u32_u24(+xor)", 8), 4: ("v_addc_co_u32", 8), 5: ("add_u64(+shift)", 8), 6: ("v_fma_f64", 8),
         7: ("v_fma_f32", 8), 8: ("modmul secp256k1 (N=8)", 1), 9: ("modmul bls12-381 (N=12)", 1),
         10: ("modsqr bls12-381", 1), 11: ("modadd bls12-381", 1),
         13: ("modmul bls12-381 radix-2^29 lazy", 1), 14: ("modsqr bls12-381 radix-2^29 lazy", 1)}


def main():
    only = [int(x) for x in sys.argv[1:]]
    if only:
        for k in list(KINDS):
            if k not in only:
                del KINDS[k]
    eng = get_engine()
    res = {}
    CUS, CLK = 256, 2.4e9
    for kind, (name, per_iter) in KINDS.items():
        for waves_per_simd in (1, 2, 4, 8):
            blocks = CUS * waves_per_simd  # 256-thread blocks = 4 waves = 1 wave/SIMD each
            iters = 4000 if kind < 8 or kind == 12 else (400 if kind != 11 else 4000)
            ms = min(eng.ubench(kind, blocks, 256, iters) for _ in range(3))
            ops = blocks * 256 * iters * per_iter
            rate = ops / (ms * 1e-3)
            # lane-ops per CU per clock (at 2.4 GHz nominal)
            per_cu_clk = rate / CUS / CLK
            res.setdefault(name, {})[waves_per_simd] = dict(ms=ms, Gops=rate / 1e9, lanes_per_cu_clk=per_cu_clk)
            print("%-28s waves/SIMD=%d  %8.3f ms  %10.1f Gop/s  %6.2f lane-ops/CU/clk" % (
                name, waves_per_simd, ms, rate / 1e9, per_cu_clk), flush=True)
    with open("gpurun_out/ubench.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    main()
