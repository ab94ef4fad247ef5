#!/bin/bash
val() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2; do
echo -n "unfused 64, L3 aligned: "; val --path unfused --frames 64 --steps 3 --warmup 1
echo -n "unfused 64, L3 round-1: "; SRCNN_DEBUG_L3=0 val --path unfused --frames 64 --steps 3 --warmup 1
echo -n "unfused 1, L3 aligned: "; val --path unfused --steps 20
echo -n "unfused 1, L3 round-1: "; SRCNN_DEBUG_L3=0 val --path unfused --steps 20
done
