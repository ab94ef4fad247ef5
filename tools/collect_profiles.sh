#!/bin/bash
# Copy the judged summaries of a tools/profile_round.sh run (scratch: gpurun_out/...) into profiles/rNN/ (tracked).
#   tools/collect_profiles.sh gpurun_out/prof5 profiles/r05
SRC=${1:?scratch directory of profile_round.sh}; DST=${2:?profiles/rNN}
mkdir -p $DST
for f in bench_default.json bench_driver_line.json bench_driver_line_no_deferral.json bench_refbytes.json bench_refbytes_b64.json bench_refbytes16.json \
         bench_split16.json bench_split16_b64.json measurements.jsonl other_kernels_stats.csv cli_process_cold.txt clock_ramp.txt diag_light_4k.txt \
         diag_light_8k.txt soak.txt soak_paths.txt soak_models.txt parity_stats_4k.txt parity_stats_split16_4k.txt stripe_overhead.txt \
         stripe_projection.txt stripe_projection_no_deferral.txt stripe_projection_refbytes.txt unfused_4k_pmc_summary.json \
         split16_4k_pmc_summary.json split16_4k_kernel_stats.csv; do
  [ -f $SRC/$f ] && grep -v "amdgpu.ids" $SRC/$f > $DST/$f
done
[ -f $SRC/mfma_4k_pmc_summary.json ] && cp $SRC/mfma_4k_pmc_summary.json $DST/fused_4k_pmc_summary.json
[ -f $SRC/mfma_4k_kernel_stats.csv ] && cp $SRC/mfma_4k_kernel_stats.csv $DST/fused_4k_kernel_stats.csv
[ -f $SRC/fix_apply_ab_final.txt ] && grep -v "amdgpu.ids" $SRC/fix_apply_ab_final.txt > $DST/fix_apply_ab_final.txt
[ -f $SRC/seam_deferral_ab_final.txt ] && grep -v "amdgpu.ids" $SRC/seam_deferral_ab_final.txt > $DST/seam_deferral_ab_final.txt
for f in $SRC/trace_refbytes/*kernel_stats.csv; do [ -f "$f" ] && cp $f $DST/refbytes_4k_kernel_stats.csv; done
for f in $SRC/trace_refbytes16/*kernel_stats.csv; do [ -f "$f" ] && cp $f $DST/refbytes16_4k_kernel_stats.csv; done
[ -f $SRC/pmc_traffic.json ] && cp $SRC/pmc_traffic.json profiles/pmc_traffic.json
ls $DST
