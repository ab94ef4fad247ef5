#!/bin/bash
# ablations of the fused kernel (SRCNN_ABLATION_BUILD library), size sweep, PMC calibration for dword accesses
OUT=gpurun_out/r02b
ROOT=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
{
echo "# abl bits: 1 no row barrier, 2 no layer-3 vertical/horizontal sums, 4 no ReLU/bias VALU, 8 no Y staging, 16 B operands from a register"
for A in 0 1 2 4 8 16 6 7 15 31 0; do echo -n "abl=$A 1x3840x2160: "; SRCNN_DEBUG_TUNE=$((A*256)) kms --steps 30; done
for A in 0 1 7 31; do echo -n "abl=$A 16x3840x2160: "; SRCNN_DEBUG_TUNE=$((A*256)) kms --steps 4 --frames 16; done
echo "# size sweep (production kernel): kernel_ms frac"
for WH in "3840 540" "3840 1080" "3840 2160" "3840 4320" "3840 8640" "7680 4320" "1920 1080" "1920 2160" "1920 4320"; do set -- $WH; echo -n "$1x$2: "; kms --steps 20 --width $1 --height $2; done
echo "# seams off (halo recompute) / rows only"
for S in 0 1 3; do echo -n "SEAMS=$S: "; SRCNN_DEBUG_SEAMS=$S kms --steps 30; done
} > $OUT/ablation.txt 2>&1
for grp in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/calib_$grp -o c -- $ROOT/build/pmc_calib ) > $OUT/calib_$grp.log 2>&1
done
python - <<'PY' > gpurun_out/r02b/pmc_calibration.txt
import csv, glob, collections
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r02b/calib_{grp}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:40]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(grp, k, len(v), "mean KB", sum(v) / len(v), "-> x", 1048576.0 / max(1e-9, sum(v) / len(v)))
PY
cat $OUT/ablation.txt $OUT/pmc_calibration.txt
