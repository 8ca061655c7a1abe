#!/bin/bash
for c in 13 14 15 16; do echo "NCG_MSM_C=$c"; NCG_MSM_C=$c timeout 300 python tools/_scratch/msm_timing.py 2>&1 | grep "^curve"; done
