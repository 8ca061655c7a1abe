#!/bin/bash
# tools/ab_two_libs.sh <libA> <libB> : share_ab.py (whole MSM on a resident set + G = 8 share) alternating two builds, G1 and G2
for rep in 1 2; do for curve in g1 g2; do for lib in "$@"; do
  NCG_LIB=$PWD/$lib timeout 200 python tools/share_ab.py --curve $curve --tag "$(basename $lib)" 2>/dev/null | tail -1
done; done; done
