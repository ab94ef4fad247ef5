#!/bin/bash
ROOT=$(pwd); OUT=gpurun_out/r02k; mkdir -p $OUT; export TMPDIR=/tmp
for PAD in 0 64 1088 4160 16448 65600; do
  ( cd /tmp && SRCNN_DEBUG_PLPAD=$PAD rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/t_$PAD -o trace -- \
      python3 $ROOT/bench.py --path unfused --frames 4 --steps 6 --warmup 2 --prewarm-ms 300 --no-cpu-baseline ) > $OUT/t_$PAD.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/t_$PAD/**/*kernel_stats.csv", recursive=True):
    for r in csv.reader(open(f)):
        if "strip_kernel" in r[0]: print("pad $PAD floats:", r[0][19:48], "avg us per frame %.1f" % (float(r[3]) / 4000))
PY
done
find $OUT -name "*kernel_trace.csv" -delete
