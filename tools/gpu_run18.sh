#!/bin/bash
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for R in "6.85,8.35,4.3" "6.85,8.05,4.3" "6.85,7.9,4.3" "6.85,8.2,4.3" "6.9,8.05,4.3" "6.85,8.05,5.0"; do
  for i in 1 2; do echo -n "rates $R: "; SRCNN_DEBUG_RATES=$R kms --steps 40; done
done
for R in "6.85,8.35,4.3" "6.85,8.05,4.3"; do echo -n "1080p rates $R: "; SRCNN_DEBUG_RATES=$R kms --steps 40 --width 1920 --height 1080; echo -n "8K rates $R: "; SRCNN_DEBUG_RATES=$R kms --steps 10 --width 7680 --height 4320;  echo -n "5760x3240 rates $R: "; SRCNN_DEBUG_RATES=$R kms --steps 10 --width 5760 --height 3240; done
SRCNN_DEBUG_RATES="6.85,8.05,4.3" python tools/diag_light.py 2>&1 | grep -E "half|CUs used"
