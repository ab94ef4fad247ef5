OUT=gpurun_out/r6b; mkdir -p $OUT; ROOT=$(pwd); export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench_driver_line.json
python bench.py --mode refbytes --no-cpu-baseline > $OUT/bench_refbytes.json 2>> $OUT/bench.err
for sz in "3840 2160" "1920 1080"; do set -- $sz
  D=$OUT/trace_refbytes_$1
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$D -o trace -- \
      python3 $ROOT/bench.py --mode refbytes --width $1 --height $2 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-refbytes --no-lanes ) > $D.log 2>&1
  echo "# rocprofv3 --kernel-trace --stats, bench.py --mode refbytes $1x$2"
  grep -h "srcnn\|fix_" $D/*kernel_stats.csv 2>/dev/null | grep -v probe | cut -d, -f1-4
  find $D -name "*kernel_trace.csv" -size +2M -delete
done
