#!/bin/bash
SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_e2.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_fused or alignment or full_size_4k" 2>&1 | tail -3
VARS="e2" tools/gpu_run21.sh
