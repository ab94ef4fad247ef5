#!/bin/bash
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
echo "# ablation build: 0 | 32 half as many layer-1 B-operand LDS reads | 8 no Y staging | 4 no ReLU"
for i in 1 2 3; do for A in 0 32 8 4; do echo -n "abl=$A: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_abl.so SRCNN_DEBUG_TUNE=$((A*256)) kms --steps 40; done; done
