#!/bin/bash
# needs an A/B build of the library: make -C noble-curves_amd/csrc clean all EXTRA=-DNCG_AB_BUILD (the shipped library ignores the NCG_* variant switches, csrc/knobs.hpp)
# A/B of the secp256k1 batch-multiply variants (NCG_SECP_W[:NCG_AFF_K]) on the GPU box
mkdir -p gpurun_out/ab
for v in ${*:-154 253 243 244 253:4 253:16}; do
  w=${v%%:*}; k=8; [[ $v == *:* ]] && k=${v##*:}
  NCG_SECP_W=$w NCG_AFF_K=$k timeout 200 python bench.py --workload secp256k1 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NCG_SECP_W=$w K=$k', round(d['ms_per_step'],3), 'ms', round(d['roofline']['kernel_ms'],3))"
done | tee gpurun_out/ab/secp_ab.txt
