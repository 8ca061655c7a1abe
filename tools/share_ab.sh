#!/bin/bash
# tools/share_ab.sh : the share-mode knobs against each other on one box (A/B build), G1 2^20 and G2 2^18, G = 8
export NCG_LIB=$PWD/tools/_build/libncg_ab.so
for rep in 1 2; do
for curve in g1 g2; do
  for cfg in "0 0 128" "1 0 128" "0 1 128" "1 1 128" "1 1 256" "1 1 512"; do
    set -- $cfg
    NCG_MSM_MERGE_UNITS=$1 NCG_MSM_TOTALS_SPLIT=$2 NCG_MSM_SHARE_QBLOCKS=$3 timeout 200 python tools/share_ab.py --curve $curve --tag "units=$1 split=$2 q=$3" 2>/dev/null | tail -1
  done
done
done
