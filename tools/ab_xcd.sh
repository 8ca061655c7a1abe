#!/bin/bash
# needs an A/B build of the library: make -C noble-curves_amd/csrc clean all EXTRA=-DNCG_AB_BUILD (the shipped library ignores the NCG_* variant switches, csrc/knobs.hpp)
# A/B of the XCD-aware block mapping of the MSM sort kernels (NCG_MSM_XCD=0/1) on one box
for rep in 1 2; do for x in 0 1; do
  for w in msm_g1 msm_g2; do
    NCG_MSM_XCD=$x timeout 300 python bench.py --workload $w --no-cpu-baseline --no-live-pmc --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xcd=$x', '$w', round(d['ms_per_msm'],3), round(d['resident_subgroup_set']['ms_per_msm'],3))"
  done
done; done
