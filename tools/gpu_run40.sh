#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tests/checks/soak_paths.py 120 13 2>&1 | tail -2
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['output_crc32'][0])"; }
for i in 1 2; do
echo -n "4K merged: "; kms --steps 50
echo -n "4K two launches, old plan: "; SRCNN_DEBUG_SEPARATE=0 kms --steps 50
echo -n "1080p merged: "; kms --steps 100 --width 1920 --height 1080
echo -n "1080p old plan: "; SRCNN_DEBUG_SEPARATE=0 kms --steps 100 --width 1920 --height 1080
echo -n "8K merged: "; kms --steps 10 --width 7680 --height 4320
echo -n "8K old plan: "; SRCNN_DEBUG_SEPARATE=0 kms --steps 10 --width 7680 --height 4320
echo -n "5760x3240x8 merged: "; kms --steps 5 --width 5760 --height 3240 --frames 8
echo -n "5760x3240x8 old plan: "; SRCNN_DEBUG_SEPARATE=0 kms --steps 5 --width 5760 --height 3240 --frames 8
done
