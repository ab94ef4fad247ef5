#!/bin/bash
# non-temporal plane loads (Convolution55 alone, EXP 4) / stores (Convolution99x11 alone, EXP 8) / both (12): same-box A/B
echo "# unfused 8 frames"; VARS="e4 e8 e12" tools/gpu_run21.sh --path unfused --frames 8 --steps 6 --warmup 5
echo "# unfused 1 frame"; VARS="e4 e8 e12" tools/gpu_run21.sh --path unfused --frames 1 --steps 20 --warmup 20 | head -8
