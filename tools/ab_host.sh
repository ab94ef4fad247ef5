# same-box A/B of the host frame stream (bench.py --path host) against a variant library: tools/ab_host.sh VARIANT
V=$1
for i in 1 2 3; do for lib in product $V; do L=""; [ $lib != product ] && L="--lib $(pwd)/srcnn_cpp_amd/libsrcnn_amd_$lib.so"
 echo -n "$lib: "; python bench.py --no-cpu-baseline --no-e2e --no-refbytes --no-lanes --path host --frames 32 --steps 4 $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done; done
