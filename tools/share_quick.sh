#!/bin/bash
# tools/share_quick.sh : one rank's share of the window-sharded MSM at G = 8 on a resident set (tools/msm_share.py), G1 2^20 and G2 2^18: latency form and in-flight form
for cfg in "g1 20" "g2 18"; do set -- $cfg
timeout 600 python tools/msm_share.py --curve $1 --log2n $2 --parts 8 --reps 8 2>/dev/null | grep -E "^(resident) " | python -c "
import sys, json
for l in sys.stdin:
    k, j = l.split(' ', 1); d = json.loads(j); s = d['share8']
    print('$1', k, 'one GPU', d['sync']['median_ms'], 'share8', {x: v for x, v in s.items() if not isinstance(v, dict)}, 'local', s['local_part0_sync']['median_ms'], 'combine+finish', s['combine_finish']['median_ms'])
"; done
