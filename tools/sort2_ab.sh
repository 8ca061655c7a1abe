#!/bin/bash
# tools/sort2_ab.sh : one-level against two-level counting sort on one box (A/B build): whole MSMs and the G = 8 share
export NCG_LIB=$PWD/tools/_build/libncg_ab.so
for rep in 1 2; do
for curve in g1 g2; do
  for s2 in 0 1; do
    NCG_MSM_SORT2=$s2 timeout 200 python tools/share_ab.py --curve $curve --tag "sort2=$s2" 2>/dev/null | tail -1
  done
done
done
