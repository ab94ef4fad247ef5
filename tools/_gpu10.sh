OUT=gpurun_out/r6j; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
for cfg in "0 0" "1 0" "2 0" "1 1" "2 1" "0 0" "1 0" "2 0"; do set -- $cfg; echo "# DRAW=$1 LDS=$2"; SRCNN_DEBUG_FIX_DRAW=$1 SRCNN_DEBUG_FIX_LDS=$2 python tools/ab_refbytes.py --lib $ROOT/srcnn_cpp_amd/libsrcnn_amd_tuning.so --sizes 3840x2160,1920x1080,1280x720,7680x4320 --margins 4 --modes refbytes --locals 0.3875 2>&1 | grep -v "strict 0\|amdgpu.ids\|library\|mfma  "; done > $OUT/fix_draw_ab.txt 2>&1
cat $OUT/fix_draw_ab.txt | cut -c1-110
