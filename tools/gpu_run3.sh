#!/bin/bash
OUT=gpurun_out/r02c
mkdir -p $OUT
python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_configs.py > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for i in 1 2 3; do echo -n "1x3840x2160: "; kms --steps 30; done
echo -n "16x: "; kms --steps 4 --frames 16
echo -n "unfused 8x: "; kms --steps 4 --frames 8 --path unfused
echo -n "7680x4320: "; kms --steps 10 --width 7680 --height 4320
