OUT=gpurun_out/r6h; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
for cfg in "0 0" "1 0" "0 1" "1 1" "0 0" "1 0"; do set -- $cfg; echo "# SPREAD=$1 LDS=$2"; SRCNN_DEBUG_FIX_SPREAD=$1 SRCNN_DEBUG_FIX_LDS=$2 python tools/ab_refbytes.py --lib $ROOT/srcnn_cpp_amd/libsrcnn_amd_tuning.so --sizes 3840x2160,1920x1080,1280x720,576x576 --margins 4 --modes refbytes --locals 0.3875 2>&1 | grep -v "strict 0\|amdgpu.ids\|library"; done > $OUT/fix_spread_ab.txt 2>&1
cat $OUT/fix_spread_ab.txt
