#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['ms_per_step'])"; }
for i in 1 2; do
echo -n "1 frame: "; kms --steps 30
echo -n "8 frames (items per frame): "; kms --steps 6 --frames 8
echo -n "8 frames (regular grid, TUNE=128): "; SRCNN_DEBUG_TUNE=128 kms --steps 6 --frames 8
echo -n "64 frames (items): "; kms --steps 3 --frames 64 --warmup 1
echo -n "64 frames (regular grid): "; SRCNN_DEBUG_TUNE=128 kms --steps 3 --frames 64 --warmup 1
done
echo -n "8 x 5760x3240: "; kms --steps 4 --frames 8 --width 5760 --height 3240
echo -n "8 x 5760x3240 regular: "; SRCNN_DEBUG_TUNE=128 kms --steps 4 --frames 8 --width 5760 --height 3240
