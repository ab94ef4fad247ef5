#!/bin/bash
OUT=gpurun_out/r02e
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for i in 1 2; do echo -n "balanced 1x3840x2160: "; kms --steps 30; done
for i in 1 2; do echo -n "old plan 1x3840x2160: "; SRCNN_DEBUG_PLAN=1 kms --steps 30; done
for K in 6 8 12 14; do echo -n "skew $K: "; SRCNN_DEBUG_SKEW=$K kms --steps 30; done
echo -n "7680x4320: "; kms --steps 10 --width 7680 --height 4320
echo -n "1920x1080: "; kms --steps 30 --width 1920 --height 1080
echo -n "5760x3240: "; kms --steps 10 --width 5760 --height 3240
echo -n "576x576: "; kms --steps 30 --width 576 --height 576
python tools/diag_light.py
