#!/bin/bash
# Experiment (tuning build): may shorter work items keep the seam windows of neighbouring strips apart (one seam launch, and with
# seam deferral none)?  SRCNN_DEBUG_SEP_MINROWS = least item height for which the planner tries (product: 24), SRCNN_DEBUG_SEP_SAVED =
# microseconds of balance a separated plan may cost (product: 2).   tools/ab_plan_sep.sh > profiles/rNN/plan_separation_ab.txt
export TMPDIR=/tmp
LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_tuning.so
for sz in "1280 720" "1920 1080" "960 540" "1600 900" "2560 1440" "7680 540" "3840 2160"; do set -- $sz
  for cfg in "24 2" "12 2" "12 8" "24 8"; do set -- $sz $cfg
    for d in on off; do
      echo -n "$1x$2 min_rows $3 saved $4 deferral $d: "
      SRCNN_DEBUG_SEP_MINROWS=$3 SRCNN_DEBUG_SEP_SAVED=$4 python bench.py --lib $LIB --no-cpu-baseline --no-e2e --no-refbytes --sustained-s 0 --seam-deferral $d --width $1 --height $2 --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['config']['output_crc32'])"
    done
  done
done
