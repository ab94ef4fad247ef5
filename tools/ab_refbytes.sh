export TMPDIR=/tmp
python -m pytest tests/test_gpu_refbytes.py -m gpu -x -q 2>&1 | tail -4
for sz in "3840 2160" "1920 1080" "7680 4320"; do set -- $sz
 for i in 1 2; do
  for lib in product v1; do
   L=""; [ $lib = v1 ] && L="--lib $(pwd)/srcnn_cpp_amd/libsrcnn_amd_v1.so"
   echo -n "$1x$2 refbytes $lib: "; python bench.py --no-cpu-baseline --no-e2e --no-refbytes --mode refbytes --width $1 --height $2 --steps 40 $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
 done
 echo -n "$1x$2 mfma: "; python bench.py --no-cpu-baseline --no-e2e --no-refbytes --width $1 --height $2 --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done
