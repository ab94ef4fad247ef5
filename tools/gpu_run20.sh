#!/bin/bash
export SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_abl.so
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
echo "# current kernel (ablation build): bits 2 all layer-3 sums, 64 no column-seam export, 128 no horizontal sums/stores, 256 no layer-3 chains"
for A in 0 64 128 256 192 2 0 64 128 256; do echo -n "abl=$A: "; SRCNN_DEBUG_TUNE=$((A*256)) kms --steps 40; done
