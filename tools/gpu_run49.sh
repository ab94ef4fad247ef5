#!/bin/bash
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['output_crc32'][0])"; }
for i in 1 2 3; do echo -n "exact (SLP packs layer 2): "; kms --mode exact --steps 5; echo -n "exact -fno-slp-vectorize:  "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_xs.so kms --mode exact --steps 5; done
