#!/bin/bash
# tools/probe_variants.sh : on whatever box this call landed, alternate the shipped library with the variants in tools/_build
# (libncg_pf.so = LDS-DMA point prefetch in the accumulate kernel, libncg_ool.so = out-of-line field multiply in the MSM kernels)
# on the two MSMs.  Fast boxes read about 3.4 (G2) / 3.6 (G1) ms with the shipped build, the slow kind 4.4 / 4.0.
rocm-smi --showserial --showbus 2>/dev/null | grep -E "Serial|PCI Bus"   # which GPU of the pool this call got
libs="noble-curves_amd/libncg.so"
for v in pf ool; do [ -f tools/_build/libncg_$v.so ] && libs="$libs tools/_build/libncg_$v.so"; done
bash tools/ab_lib.sh msm_g2 $libs
bash tools/ab_lib.sh msm_g1 $libs
