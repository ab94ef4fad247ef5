# timing-only experiment: the FAST row body without its 12 chain adds (wrong pixels) against the product -- the upper bound of
# what layer-3 accumulation in place could save.  Variant: SRCNN_BUILD_VARIANT=nochain SRCNN_BUILD_DEFINES=-DSRCNN_ABL_NO_CHAIN_ADDS
kms() { python bench.py --no-cpu-baseline --no-e2e --no-refbytes --no-lanes "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for i in 1 2 3; do
  echo -n "product: "; kms --steps 60
  echo -n "no chain adds: "; kms --steps 60 --lib $(pwd)/srcnn_cpp_amd/libsrcnn_amd_nochain.so
done
