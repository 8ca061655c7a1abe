#!/bin/bash
# needs an A/B build of the library: make -C noble-curves_amd/csrc clean all EXTRA=-DNCG_AB_BUILD (the shipped library ignores the NCG_* variant switches, csrc/knobs.hpp)
# A/B of two libncg builds on the MSM workloads with per-kernel times: tools/ab_msm.sh <tag> <lib...>
TAG=$1; shift
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  (cd /tmp && NCG_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/st_$name -- python $GRAFT_REPO_ROOT/tools/msm_timing.py > $GRAFT_REPO_ROOT/gpurun_out/$TAG/st_$name.log 2>&1)
  grep "^curve" gpurun_out/$TAG/st_$name.log
  python tools/sum_stats.py gpurun_out/$TAG/st_$name | grep -v "digits\|hist\|scan\|totals\|to_mont\|gen\|fixup_write"
  find gpurun_out/$TAG/st_$name -name "*.db" -delete; find gpurun_out/$TAG/st_$name -name "*trace.csv" -size +2M -delete
done
