#!/bin/bash
# tools/ab_small_msm.sh <tag> : the small-plan MSM variants side by side on one box (A/B builds from tools/build_variant.sh):
# merge tree of 4 units per bucket, of 2, and one unit per bucket (NCG_MSM_MERGE_TREE=0); accumulate CU spreading off.
TAG=${1:-ab_small}
mkdir -p gpurun_out/$TAG
run() { name=$1; shift; env "$@" timeout 600 python tools/msm_timing.py --min 10 --max 16 --reps 15 --no-resident 2>/dev/null | grep "^{" | python -c "
import sys, json
print('$name', ' '.join('%s%d:%.3f' % (r['curve'], r['log2n'], r['generic']['median_ms']) for r in map(json.loads, sys.stdin)))" | tee -a gpurun_out/$TAG/ab.txt; }
for rep in 1 2; do
run tree_u4   NCG_LIB=$GRAFT_REPO_ROOT/tools/_build/libncg_u2.so
run tree_u2   NCG_LIB=$GRAFT_REPO_ROOT/tools/_build/libncg_u1.so
run serial15  NCG_LIB=$GRAFT_REPO_ROOT/tools/_build/libncg_u2.so NCG_MSM_MERGE_TREE=0
run nospread  NCG_LIB=$GRAFT_REPO_ROOT/tools/_build/libncg_u2.so NCG_MSM_ACCUM_SPREAD=0
run tail2     NCG_LIB=$GRAFT_REPO_ROOT/tools/_build/libncg_u2.so NCG_MSM_TAIL_ROUNDS=2
run nofused   NCG_LIB=$GRAFT_REPO_ROOT/tools/_build/libncg_u2.so NCG_MSM_SMALL_SORT=0
done
