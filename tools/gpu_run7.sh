#!/bin/bash
OUT=gpurun_out/r02g
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for i in 1 2 3; do echo -n "calibrated 1x3840x2160: "; kms --steps 30; done
for i in 1 2; do echo -n "balanced, no calibration: "; SRCNN_DEBUG_CALIB=0 kms --steps 30; done
for i in 1 2; do echo -n "old plan: "; SRCNN_DEBUG_PLAN=1 kms --steps 30; done
echo -n "7680x4320: "; kms --steps 10 --width 7680 --height 4320
echo -n "7680x4320 old: "; SRCNN_DEBUG_PLAN=1 kms --steps 10 --width 7680 --height 4320
echo -n "1920x1080: "; kms --steps 30 --width 1920 --height 1080
echo -n "1920x1080 old: "; SRCNN_DEBUG_PLAN=1 kms --steps 30 --width 1920 --height 1080
echo -n "5760x3240: "; kms --steps 10 --width 5760 --height 3240
echo -n "5760x3240 old: "; SRCNN_DEBUG_PLAN=1 kms --steps 10 --width 5760 --height 3240
echo "=== calibrated"; python tools/diag_light.py 2>&1 | grep -v amdgpu.ids | grep -v "^  *[0-9]"
echo "=== no calib"; SRCNN_DEBUG_CALIB=0 python tools/diag_light.py 2>&1 | grep -v amdgpu.ids | grep -v "^  *[0-9]"
