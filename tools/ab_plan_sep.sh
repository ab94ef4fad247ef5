#!/bin/bash
# Experiment (tuning build): may shorter work items keep the seam windows of neighbouring strips apart (one seam launch, and with
# seam deferral none)?  SRCNN_DEBUG_SEP_MINROWS = least item height for which the planner tries, SRCNN_DEBUG_SEP_SAVED =
# microseconds of balance a separated plan may cost.  Configurations alternate A B B A per size so that clock drift over the
# minutes of a run does not read as a difference.   tools/ab_plan_sep.sh > profiles/rNN/plan_separation_ab.txt
export TMPDIR=/tmp
LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_tuning.so
A=${A:-"24 2"}; B=${B:-"12 8"}
for sz in "1280 720" "1366 768" "1440 900" "1600 900" "1280 1024" "1920 1080" "1920 1200" "2048 1152" "2560 1440" "3840 2160"; do set -- $sz
  for cfg in "$A" "$B" "$B" "$A"; do set -- $sz $cfg
    for d in on off; do
      echo -n "$1x$2 min_rows $3 saved $4 deferral $d: "
      SRCNN_DEBUG_PLANLOG=1 SRCNN_DEBUG_SEP_MINROWS=$3 SRCNN_DEBUG_SEP_SAVED=$4 python bench.py --lib $LIB --no-cpu-baseline --no-e2e --no-refbytes --no-lanes --sustained-s 0 --seam-deferral $d --width $1 --height $2 --steps 40 2>/tmp/planlog.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['config']['output_crc32'], end=' ')"
      grep -m1 "^plan:" /tmp/planlog.txt || echo
    done
  done
done
