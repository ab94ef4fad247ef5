#!/bin/bash
# same-box A/B of two library builds: tools/ab.sh VARIANT [bench args...]   (libsrcnn_amd_VARIANT.so vs libsrcnn_amd.so)
V=$1; shift
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for i in 1 2 3; do
  echo -n "product: "; kms --warmup 300 --steps 50 "$@"
  echo -n "$V: "; SRCNN_LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_$V.so kms --warmup 300 --steps 50 "$@"
done
