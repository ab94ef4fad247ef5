#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_split16.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
VARS="e1" tools/gpu_run21.sh
