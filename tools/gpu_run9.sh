#!/bin/bash
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['ms_per_step'])"; }
echo -n "full: "; kms --steps 30
echo -n "strip only: "; SRCNN_DEBUG_TUNE=32 kms --steps 30
echo -n "empty strip launch only (TUNE=96): "; SRCNN_DEBUG_TUNE=96 kms --steps 30
echo -n "empty strip launch + seam kernels (TUNE=64): "; SRCNN_DEBUG_TUNE=64 kms --steps 30
python tools/diag_light.py 2>&1 | grep -E "kernel span"
