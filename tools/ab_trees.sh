#!/bin/bash
# tools/ab_trees.sh <out-dir> <treeA> <treeB> [reps] : tools/bench_extra.py of two whole trees (each with its own libncg.so and Python side),
# alternating on ONE box - tells a box-to-box difference from a code difference in the side table of BASELINE.md.
OUT=$1; A=$2; B=$3; REPS=${4:-2}
mkdir -p $OUT
for rep in $(seq 1 $REPS); do for t in $A $B; do
  tag=$(basename $t)
  (cd $t && timeout 600 python tools/bench_extra.py --out /tmp/extra_$tag.json > /dev/null 2>$GRAFT_REPO_ROOT/$OUT/extra_${tag}_$rep.err)
  cp /tmp/extra_$tag.json $OUT/extra_${tag}_$rep.json 2>/dev/null
done; done
ls $OUT
