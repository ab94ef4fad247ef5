#!/bin/bash
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for K in 10 9 8 7 6; do
  for i in 1 2; do echo -n "skew $K rates 6.85,8.05: "; SRCNN_DEBUG_SKEW=$K SRCNN_DEBUG_RATES="6.85,8.05,4.3" kms --steps 40; done
done
for K in 10 8; do echo -n "1080p skew $K: "; SRCNN_DEBUG_SKEW=$K SRCNN_DEBUG_RATES="6.85,8.05,4.3" kms --steps 40 --width 1920 --height 1080; echo -n "8K skew $K: "; SRCNN_DEBUG_SKEW=$K SRCNN_DEBUG_RATES="6.85,8.05,4.3" kms --steps 10 --width 7680 --height 4320; done
SRCNN_DEBUG_SKEW=8 SRCNN_DEBUG_RATES="6.85,8.05,4.3" python tools/diag_light.py 2>&1 | grep -E "half|CUs used"
