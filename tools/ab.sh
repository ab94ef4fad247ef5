#!/bin/bash
# same-box A/B of two library builds: tools/ab.sh VARIANT [bench args...]   (libsrcnn_amd_VARIANT.so vs libsrcnn_amd.so;
# build the variant with SRCNN_BUILD_VARIANT=name SRCNN_BUILD_DEFINES="-D..." python -m srcnn_cpp_amd.build).
# Prints kernel ms per step, fraction of the f32-MFMA peak and the output crc32, alternating the two builds REPS times.
V=$1; shift
REPS=${REPS:-3}
kms() { python bench.py --no-cpu-baseline --no-e2e --no-refbytes --no-lanes "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['output_crc32'][0])"; }
for i in $(seq $REPS); do
  echo -n "product: "; kms --steps 50 "$@"
  echo -n "$V: "; kms --lib $(pwd)/srcnn_cpp_amd/libsrcnn_amd_$V.so --steps 50 "$@"
done
