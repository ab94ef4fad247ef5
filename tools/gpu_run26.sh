#!/bin/bash
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
VARS="e1" tools/gpu_run21.sh
echo "# 1080p"; VARS="e1" tools/gpu_run21.sh --width 1920 --height 1080 | head -4
echo "# 8 frames"; VARS="e1" tools/gpu_run21.sh --frames 8 --steps 10 | head -4
