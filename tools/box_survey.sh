#!/bin/bash
# tools/box_survey.sh : one line per box - GPU serial and the five workloads' times of bench.py (no CPU baseline, no counter passes);
# called once per gpurun call (every call lands on whatever box the pool hands out): the spread across boxes for ONE build.
ser=$(rocm-smi --showserial 2>/dev/null | grep -o "Serial Number: .*" | head -1 | awk '{print $3}')
timeout 400 python bench.py --no-cpu-baseline --no-live-pmc --quick-verify --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('serial $ser secp256k1 %.3f ms  G1 MSM %.3f  G2 MSM %.3f  ed25519 %.3f  NTT %.4f' % (d['ms_per_step'], d['msm_g1']['ms_per_msm'], d['msm_g2']['ms_per_msm'], d['ed25519']['ms_per_batch'], d['ntt']['ms_per_transform']))"
