#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -3
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['output_crc32'][0])"; }
for i in 1 2 3; do
echo -n "4K merged: "; kms --steps 50
echo -n "4K two launches: "; SRCNN_DEBUG_SEAM_MERGE=0 kms --steps 50
echo -n "4K old plan + two launches: "; SRCNN_DEBUG_SEPARATE=0 kms --steps 50
done
for i in 1 2; do
echo -n "1080p merged: "; kms --steps 100 --width 1920 --height 1080
echo -n "1080p two launches: "; SRCNN_DEBUG_SEAM_MERGE=0 kms --steps 100 --width 1920 --height 1080
echo -n "1080p old plan: "; SRCNN_DEBUG_SEPARATE=0 kms --steps 100 --width 1920 --height 1080
echo -n "8x4K merged: "; kms --steps 10 --frames 8
echo -n "8x4K old plan: "; SRCNN_DEBUG_SEPARATE=0 kms --steps 10 --frames 8
done
