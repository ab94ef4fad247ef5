#!/bin/bash
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
echo "# unfused 8 frames"; VARS="e1" tools/gpu_run21.sh --path unfused --frames 8 --steps 6 --warmup 5
echo "# unfused 1 frame"; VARS="e1" tools/gpu_run21.sh --path unfused --frames 1 --steps 20 --warmup 20 | head -4
