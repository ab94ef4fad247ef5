#!/bin/bash
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for i in 1 2; do echo -n "3 kernels: "; kms --steps 30; done
for i in 1 2; do echo -n "strip kernel only (TUNE=32): "; SRCNN_DEBUG_TUNE=32 kms --steps 30; done
for i in 1 2; do echo -n "1080p 3 kernels: "; kms --steps 30 --width 1920 --height 1080; done
for i in 1 2; do echo -n "1080p strip only: "; SRCNN_DEBUG_TUNE=32 kms --steps 30 --width 1920 --height 1080; done
