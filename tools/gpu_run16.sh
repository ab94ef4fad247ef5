#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
tools/ab.sh prev
tools/ab.sh prev --path unfused --frames 8 --steps 6 --warmup 5
tools/ab.sh prev --width 1920 --height 1080
