#!/bin/bash
SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_e2.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
echo "# 4K"; VARS="e2" tools/gpu_run21.sh | head -6
echo "# 1080p"; VARS="e2" tools/gpu_run21.sh --width 1920 --height 1080 --steps 100 | head -6
echo "# 576"; VARS="e2" tools/gpu_run21.sh --width 576 --height 576 --steps 200 | head -4
echo "# 1280x720"; VARS="e2" tools/gpu_run21.sh --width 1280 --height 720 --steps 100 | head -4
