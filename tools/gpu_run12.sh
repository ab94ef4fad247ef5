#!/bin/bash
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size or forward or host" 2>&1 | tail -3
val() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for i in 1 2; do
echo -n "host banded (4): "; val --path host --steps 20
echo -n "host one launch: "; SRCNN_DEBUG_BANDS=1 val --path host --steps 20
echo -n "host 2 bands: "; SRCNN_DEBUG_BANDS=2 val --path host --steps 20
echo -n "host 8 bands: "; SRCNN_DEBUG_BANDS=8 val --path host --steps 20
done
echo -n "host 7680x4320 banded: "; val --path host --steps 10 --width 7680 --height 4320
echo -n "host 7680x4320 one: "; SRCNN_DEBUG_BANDS=1 val --path host --steps 10 --width 7680 --height 4320
