#!/bin/bash
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['value'])"; }
echo -n "1st (fresh box, default prewarm 400 ms): "; kms
echo -n "2nd: "; kms
echo -n "3rd prewarm 3000: "; kms --prewarm-ms 3000
echo -n "4th default: "; kms
echo -n "5th prewarm 0: "; kms --prewarm-ms 0
echo -n "6th default, steps 20 warmup 5 (contract): "; kms --steps 20 --warmup 5
