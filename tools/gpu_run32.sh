#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['plan'])"; }
for i in 1 2; do
echo -n "576x576 product: "; kms --width 576 --height 576 --steps 200
echo -n "576x576 e1: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_e1.so kms --width 576 --height 576 --steps 200
echo -n "960x540 product: "; kms --width 960 --height 540 --steps 200
echo -n "960x540 e1: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_e1.so kms --width 960 --height 540 --steps 200
done
