#!/bin/bash
# per-CU (rows of the first-dispatched workgroup, rows of the one that joined it, their loop-start and exit times) of the stamped
# production launch under several row splits (SRCNN_DEBUG_RATES moves the planner's split), for fitting the planner's model
# (tools/plan_fit.py):   tools/plan_fit_collect.sh OUT.txt
OUT=${1:-gpurun_out/plan_fit.txt}; : > $OUT
export TMPDIR=/tmp
for sz in ${SIZES:-"3840 2160" "1920 1080" "7680 540" "2560 1440"}; do
  for r in ${RATES:-"6.40,8.40,3.76" "6.85,8.35,4.3" "6.3,8.0,4.3" "7.0,7.3,4.3" "6.0,8.6,4.3" "6.6,8.0,4.0" "7.3,7.3,4.3" "6.4,8.4,4.6" "6.4,9.2,3.76" "6.4,7.8,3.76"}; do
    DIAG_DUMP=$OUT SRCNN_DEBUG_RATES=$r python tools/diag_light.py $sz > /dev/null 2>&1
  done
done
wc -l $OUT
