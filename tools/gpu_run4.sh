#!/bin/bash
OUT=gpurun_out/r02d
ROOT=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
python tools/diag_light.py > $OUT/diag_light_4k.txt 2>&1
python tools/diag_light.py 3840 4320 > $OUT/diag_light_3840x4320.txt 2>&1
python tools/diag_light.py 1920 1080 > $OUT/diag_light_1080p.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/trace -o trace -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $OUT/trace.log 2>&1
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/r02d/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "srcnn" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[-12:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1000 if prev_end else 0
    print(f"{r['Kernel_Name'][12:40]:30s} dur {(e - s) / 1000:9.2f} us   gap before {gap:7.2f} us")
    prev_end = e
PY
cat $OUT/diag_light_4k.txt $OUT/diag_light_3840x4320.txt $OUT/diag_light_1080p.txt
