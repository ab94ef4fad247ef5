#!/bin/bash
# Same-box A/B of the REFBYTES fix-up: the product library against another build (default: libsrcnn_amd_r4.so = round 4's
# library, built from that commit in a worktree with SRCNN_BUILD_VARIANT=r4), then rocprofv3 per-kernel times of both at the
# product's threshold factor.
#   tools/ab_refbytes.sh [OUTFILE] [OTHER_LIB]
OUT=${1:-gpurun_out/ab_refbytes.txt}
OTHER=${2:-srcnn_cpp_amd/libsrcnn_amd_r4.so}
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $(dirname $OUT)
: > $OUT
for i in 1 2; do
  python tools/ab_refbytes.py --sizes ${SIZES:-3840x2160,1920x1080,7680x4320} >> $OUT 2>&1
  [ -f $OTHER ] && python tools/ab_refbytes.py --sizes ${SIZES:-3840x2160,1920x1080,7680x4320} --lib $ROOT/$OTHER >> $OUT 2>&1
done
for lib in product other; do
  L=""; [ $lib = other ] && { [ -f $OTHER ] || continue; L="--lib $ROOT/$OTHER --no-fix-strict"; }
  for sz in "3840 2160" "1920 1080"; do set -- $sz
    D=gpurun_out/trace_refbytes_${lib}_$1
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$D -o trace -- \
        python3 $ROOT/bench.py --mode refbytes --fix-margin 4 $L --width $1 --height $2 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-refbytes --no-lanes ) > $D.log 2>&1
    echo "# rocprofv3 --kernel-trace --stats, bench.py --mode refbytes --fix-margin 4 $L $1x$2" >> $OUT
    grep -h "srcnn" $D/*kernel_stats.csv 2>/dev/null | grep -v probe | cut -d, -f1-4 >> $OUT
    find $D -name "*kernel_trace.csv" -size +2M -delete
  done
done
cat $OUT
