#!/bin/bash
# same-box A/B of the SRCNN_EXP build variants (profiles/r02/ablation.txt section 7): kernel ms / fraction, alternating
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['output_crc32'][0])"; }
VARS=${VARS:-"e1 e2 e3 e8 e11 e15"}
for i in 1 2 3; do
  echo -n "product: "; kms --steps 50 "$@"
  for V in $VARS; do echo -n "$V: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_$V.so kms --steps 50 "$@"; done
done
