#!/bin/bash
# Experiment (tuning build): does weighing the two extra feature rows at the open ends of a rank's stripe in the work-item plan pay?
# SRCNN_DEBUG_PLAN_EDGES=0 = round 4's plans.   tools/ab_plan_edges.sh > profiles/rNN/plan_edges_ab.txt
export TMPDIR=/tmp
LIB=$(pwd)/srcnn_cpp_amd/libsrcnn_amd_tuning.so
for e in 0 1 1 0; do
  echo "## SRCNN_DEBUG_PLAN_EDGES=$e"
  SRCNN_DEBUG_PLAN_EDGES=$e python tools/stripe_projection.py --lib $LIB --ns ${NS:-1,2,4,8} --steps 40 2>/dev/null | grep -v "^#  N rank\|amdgpu.ids"
done
