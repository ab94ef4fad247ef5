#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q -k "exact or bit_exact or pipeline or butterfly" 2>&1 | tail -2
python -m pytest tests/test_abi.py -x -q 2>&1 | tail -1
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['output_crc32'][0])"; }
for i in 1 2 3; do echo -n "exact: "; kms --mode exact --steps 5; done
