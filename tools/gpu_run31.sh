#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['plan'])"; }
for i in 1 2; do
echo -n "576x576 product: "; kms --width 576 --height 576 --steps 200
echo -n "576x576 e1: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_e1.so kms --width 576 --height 576 --steps 200
echo -n "1280x720 product: "; kms --width 1280 --height 720 --steps 100
echo -n "1280x720 e1: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_e1.so kms --width 1280 --height 720 --steps 100
done
echo -n "8K product: "; kms --width 7680 --height 4320 --steps 10
echo -n "8K e1: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_e1.so kms --width 7680 --height 4320 --steps 10
echo -n "64x4K product: "; kms --frames 64 --steps 3 --warmup 2
echo -n "64x4K e1: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_e1.so kms --frames 64 --steps 3 --warmup 2
