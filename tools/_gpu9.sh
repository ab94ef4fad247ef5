OUT=gpurun_out/r6i; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
for cfg in "const 1920 1080" "synth 1920 1080" "synth 576 576" "synth 300 200" "synth 3840 2160"; do set -- $cfg
  D=$OUT/t_$1_$2
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$D -o trace -- python3 $ROOT/tools/fix_latency_probe.py $1 $2 $3 ) > $D.log 2>&1
  echo "# $cfg: $(grep flagged $D.log)"; grep -h "fix_\|seams\|strip" $D/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120; find $D -name "*kernel_trace.csv" -size +1M -delete
done
