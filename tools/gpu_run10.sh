#!/bin/bash
OUT=gpurun_out/r02j
ROOT=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for V in 1 0; do
  for F in 1 8; do
    ( cd /tmp && SRCNN_DEBUG_L3=$V rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_l3_${V}_f$F -o trace -- \
        python3 $ROOT/bench.py --path unfused --frames $F --steps 6 --warmup 2 --prewarm-ms 300 --no-cpu-baseline ) > $OUT/trace_l3_${V}_f$F.log 2>&1
    echo "L3 variant $V frames $F:"; python - <<PY
import csv, glob
for f in glob.glob("$OUT/trace_l3_${V}_f$F/**/*kernel_stats.csv", recursive=True):
    for r in csv.reader(open(f)):
        if "srcnn" in r[0]: print("   ", r[0][:60], "calls", r[1], "avg us %.1f" % (float(r[3]) / 1000))
PY
  done
done
for V in 1 0; do
  ( cd /tmp && SRCNN_DEBUG_L3=$V rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_l3_$V -o pmc -- \
      python3 $ROOT/bench.py --path unfused --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline ) > $OUT/pmc_l3_$V.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_l3_$V/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "srcnn" in row["Kernel_Name"]: acc[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
for k, v in acc.items(): print("L3 variant $V FETCH_SIZE KB", k, sum(v) / len(v), "-> x2 =", 2 * sum(v) / len(v) * 1024 / 1e6, "MB")
PY
done
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
