OUT=gpurun_out/r6g; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
python -m pytest tests/test_gpu_refbytes.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
for v in 1 0 1 0; do echo "# SRCNN_DEBUG_FIX_LDS=$v"; SRCNN_DEBUG_FIX_LDS=$v python tools/ab_refbytes.py --lib $ROOT/srcnn_cpp_amd/libsrcnn_amd_tuning.so --sizes 3840x2160,1920x1080,1280x720 --margins 4 --modes refbytes,refbytes16 --locals 0.3875 2>&1 | grep -v "strict 0\|amdgpu.ids"; done > $OUT/fix_lds_ab.txt 2>&1
cat $OUT/fix_lds_ab.txt
for v in 1 0; do D=$OUT/trace_lds$v
 ( cd /tmp && SRCNN_DEBUG_FIX_LDS=$v rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$D -o trace -- python3 $ROOT/bench.py --lib $ROOT/srcnn_cpp_amd/libsrcnn_amd_tuning.so --mode refbytes --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-refbytes --no-lanes ) > $D.log 2>&1
 echo "# LDS=$v 4K"; grep -h "fix_" $D/*kernel_stats.csv | cut -d, -f1-4; find $D -name "*kernel_trace.csv" -size +2M -delete; done
