#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['output_crc32'][:2])"; }
for F in 2 8 24; do
  echo -n "$F x 4K frame loop: "; kms --frames $F --steps 6
  echo -n "$F x 4K one launch: "; SRCNN_DEBUG_FRAMELOOP=0 kms --frames $F --steps 6
done
echo -n "8 x 5760x3240 frame loop: "; kms --frames 8 --steps 5 --width 5760 --height 3240
echo -n "8 x 5760x3240 one launch: "; SRCNN_DEBUG_FRAMELOOP=0 kms --frames 8 --steps 5 --width 5760 --height 3240
for F in 4 8 16; do
  echo -n "$F x 1080p default: "; kms --frames $F --steps 20 --width 1920 --height 1080
  echo -n "$F x 1080p one launch: "; SRCNN_DEBUG_FRAMELOOP=0 kms --frames $F --steps 20 --width 1920 --height 1080
done
