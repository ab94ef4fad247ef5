#!/bin/bash
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['output_crc32'][0])"; }
echo "# ablation build: 0 | 4 no ReLU | 516 ReLU as 24 independent single pk_mul between the layer-1 MFMAs"
for i in 1 2 3; do for A in 0 4 516; do echo -n "abl=$A: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_abl.so SRCNN_DEBUG_TUNE=$((A*256)) kms --steps 40; done; done
echo "# static priority: e1 = asm first MFMA; e17 = e1 + late workgroup prio 1; e33 = e1 + early workgroup prio 1"
for i in 1 2 3; do for V in e1 e17 e33; do echo -n "$V: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_$V.so kms --steps 50; done; done
