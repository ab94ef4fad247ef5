#!/bin/bash
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for i in 1 2 3; do
for R in "" "6.67,8.19,4.3" "6.67,8.19,4.15" "6.6,8.3,4.2"; do echo -n "rates [$R]: 4K "; SRCNN_DEBUG_RATES=$R kms --steps 50; done
done
for R in "" "6.67,8.19,4.3" "6.67,8.19,4.15"; do echo -n "rates [$R]: 1080p "; SRCNN_DEBUG_RATES=$R kms --steps 100 --width 1920 --height 1080;  echo -n "rates [$R]: 8K "; SRCNN_DEBUG_RATES=$R kms --steps 10 --width 7680 --height 4320; done
