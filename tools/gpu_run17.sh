#!/bin/bash
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
tools/ab.sh prev --path unfused --frames 8 --steps 6 --warmup 5
tools/ab.sh prev --path unfused --frames 1 --steps 20 --warmup 20
