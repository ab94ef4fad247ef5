#!/bin/bash
# A/B of library builds of the split-f16 strip kernel on one box (product = libsrcnn_amd.so, variants libsrcnn_amd_NAME.so):
#   tools/ab_s16.sh NAME [NAME ...]     ms per step and output crc32 in SPLIT16 and REFBYTES16, alternating, REPS times
export TMPDIR=/tmp
REPS=${REPS:-3}
kms() { python bench.py --no-cpu-baseline --no-e2e --no-refbytes --no-lanes --sustained-s 0 --steps 50 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['output_crc32'][0])"; }
for mode in split16 refbytes16; do
  for i in $(seq $REPS); do
    echo -n "$mode product: "; kms --mode $mode
    for v in "$@"; do echo -n "$mode $v: "; kms --mode $mode --lib $(pwd)/srcnn_cpp_amd/libsrcnn_amd_$v.so; done
  done
done
