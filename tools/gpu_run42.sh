#!/bin/bash
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for F in 2 4 8 16 24; do
  echo -n "$F x 4K items per frame: "; kms --frames $F --steps 6
  echo -n "$F x 4K regular grid:    "; SRCNN_DEBUG_TUNE=128 kms --frames $F --steps 6
done
for F in 4 16; do
  echo -n "$F x 1080p items per frame: "; kms --frames $F --steps 20 --width 1920 --height 1080
  echo -n "$F x 1080p regular grid:    "; SRCNN_DEBUG_TUNE=128 kms --frames $F --steps 20 --width 1920 --height 1080
done
