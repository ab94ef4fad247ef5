#!/bin/bash
SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_rt16.so python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -2
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['output_crc32'][0])"; }
for i in 1 2 3; do
echo -n "pipeline 32-row tiles: "; kms --path pipeline --steps 50
echo -n "pipeline 16-row tiles: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_rt16.so kms --path pipeline --steps 50
done
