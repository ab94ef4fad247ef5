#!/bin/bash
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
VARS="e1" tools/gpu_run21.sh
