#!/bin/bash
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exact or bit_exact or conv55" 2>&1 | tail -2
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['output_crc32'][0])"; }
for i in 1 2; do
echo -n "exact, scalar weights in conv55: "; kms --mode exact --steps 5
echo -n "exact, LDS weights in conv55:    "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_xl.so kms --mode exact --steps 5
done
