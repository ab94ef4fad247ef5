#!/bin/bash
# Profiles of one round on ONE MI355X (run through gpurun from the repo root):
#   tools/profile_round.sh [OUTDIR]        (default gpurun_out/prof)
#   PMC_ONLY=1 tools/profile_round.sh [OUTDIR]   only steps 2-3 (kernel trace + PMC passes + summary): what profiles/pmc_traffic.json
#                                                needs after a change of the launch code that leaves the kernels alone
# 1. the default bench line (with cpu_baseline), 2. rocprofv3 --kernel-trace --stats of the same command,
# 3. PMC passes (one counter group per run, never together with a trace), for the float32 MFMA mode and
# for the opt-in split-f16 mode, 4. the other configurations (tools/measure_all.sh), 5. in-kernel stamps.
# tools/pmc_summarize.py turns the PMC csv files into profiles/rNN/*_pmc_summary.json.
# (the profiled command lines carry --no-e2e --no-refbytes --no-lanes: only the K steps of the headline workload run under the profiler,
# so no dispatch of the host-frame pipeline or of the REFBYTES leg is averaged into the per-kernel figures)
OUT=${1:-gpurun_out/prof}
ROOT=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$PMC_ONLY" ]; then
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2>> $OUT/bench_default.err      # the driver's exact command
python bench.py --steps 20 --warmup 5 --seam-deferral off > $OUT/bench_driver_line_no_deferral.json 2>> $OUT/bench_default.err
python bench.py --mode refbytes --no-cpu-baseline > $OUT/bench_refbytes.json 2>> $OUT/bench_default.err
python bench.py --mode refbytes --frames 64 --steps 5 --no-cpu-baseline > $OUT/bench_refbytes_b64.json 2>> $OUT/bench_default.err
python bench.py --mode refbytes16 --no-cpu-baseline > $OUT/bench_refbytes16.json 2>> $OUT/bench_default.err
python bench.py --mode split16 --no-cpu-baseline > $OUT/bench_split16.json 2>> $OUT/bench_default.err
python bench.py --mode split16 --frames 64 --steps 10 --no-cpu-baseline > $OUT/bench_split16_b64.json 2>> $OUT/bench_default.err
fi
for mode in mfma split16; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_$mode -o trace -- \
      python3 $ROOT/bench.py --mode $mode --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-refbytes --no-lanes ) > $OUT/trace_$mode.log 2>&1
  for grp in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
    tag=$(echo $grp | cut -d' ' -f1)
    ( cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/pmc_${mode}_$tag -o pmc -- \
        python3 $ROOT/bench.py --mode $mode --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-refbytes --no-lanes ) > $OUT/pmc_${mode}_$tag.log 2>&1
  done
done
# the kernels beside the headline: the REFBYTES step (flag-writing strip kernel, fix_collect, fix_apply, idle
# fix_rerun) and the two pipeline byte kernels -- HBM bytes, instruction mix, issue / stall split, LDS conflicts
for run in "refbytes --mode refbytes" "pipeline --path pipeline"; do
  set -- $run; tag=$1; shift
  for grp in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
    g=$(echo $grp | cut -d' ' -f1)
    ( cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/pmc_${tag}_$g -o pmc -- \
        python3 $ROOT/bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-refbytes --no-lanes ) > $OUT/pmc_${tag}_$g.log 2>&1
  done
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_$tag -o trace -- \
      python3 $ROOT/bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-refbytes --no-lanes ) > $OUT/trace_$tag.log 2>&1
done
# ... and a kernel trace of ONE unfused 3840x2160 frame, the run the pmc_unfused_* passes below count (TB/s from the counters)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_unfused1 -o trace -- \
    python3 $ROOT/bench.py --path unfused --steps 20 --warmup 3 --no-cpu-baseline ) > $OUT/trace_unfused1.log 2>&1
if [ -z "$PMC_ONLY" ]; then
# Convolution99x11 (<1>) and Convolution55 (<2>) alone: HBM bytes (FETCH_SIZE x2, WRITE_SIZE: pmc_calibration.txt) and instruction mix
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  ( cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/pmc_unfused_$tag -o pmc -- \
      python3 $ROOT/bench.py --path unfused --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline ) > $OUT/pmc_unfused_$tag.log 2>&1
done
# the other kernels: unfused path (layer-1/2 kernel + layer-3 kernel), pipeline byte kernels, exact kernels
for cfg in "unfused --path unfused --frames 8 --steps 5" "pipeline --path pipeline --steps 20" "exact --mode exact --steps 5" "pipeline_split16 --path pipeline --mode split16 --steps 20" "refbytes --mode refbytes --steps 20" "refbytes16 --mode refbytes16 --steps 20"; do
  set -- $cfg; tag=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_$tag -o trace -- \
      python3 $ROOT/bench.py "$@" --warmup 1 --no-cpu-baseline ) > $OUT/trace_$tag.log 2>&1
done
tools/measure_all.sh $OUT/measurements.jsonl > /dev/null 2>&1
python tools/diag_light.py > $OUT/diag_light_4k.txt 2>&1
python tools/diag_light.py 7680 4320 > $OUT/diag_light_8k.txt 2>&1
python tools/evt_test.py > $OUT/clock_ramp.txt 2>&1
python tools/diag_split16.py > $OUT/diag_stamps_split16.txt 2>&1
tools/stripe_overhead.sh $OUT/stripe_overhead.txt > /dev/null 2>&1
python tools/stripe_projection.py > $OUT/stripe_projection.txt 2>&1
python tools/stripe_projection.py --seam-deferral off > $OUT/stripe_projection_no_deferral.txt 2>&1
python tools/stripe_projection.py --mode refbytes --ns 1,8 > $OUT/stripe_projection_refbytes.txt 2>&1
python tests/checks/time_cli.py 7 > $OUT/cli_process_cold.txt 2>&1
tools/ab_refbytes.sh $OUT/fix_apply_ab_final.txt > /dev/null 2>&1
tools/ab_defer.sh > $OUT/seam_deferral_ab_final.txt 2>&1
python tests/checks/soak.py 120 31 > $OUT/soak.txt 2>&1
python tests/checks/soak_paths.py 60 37 > $OUT/soak_paths.txt 2>&1
python tests/checks/soak_models.py 300 3 > $OUT/soak_models.txt 2>&1
python tools/two_lane_probe.py 3840x2160,2560x1440,1920x1080,1280x720,960x540,576x576,7680x540 > $OUT/two_lane_probe.txt 2>&1
python tests/checks/parity_stats.py > $OUT/parity_stats_4k.txt 2>&1
python tests/checks/split16_stats.py > $OUT/parity_stats_split16_4k.txt 2>&1
[ -x build/f16_probe ] && ./build/f16_probe > $OUT/f16_probe.txt 2>&1
[ -x build/mfma_probe ] && ./build/mfma_probe > $OUT/mfma_probe.txt 2>&1
[ -x build/pk_f32_probe ] && ./build/pk_f32_probe > $OUT/pk_f32_probe.txt 2>&1
fi
# keep the merge small: the raw per-dispatch csv files are summarised on the box
python tools/pmc_summarize.py $OUT > $OUT/pmc_summarize.log 2>&1
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
ls -la $OUT
