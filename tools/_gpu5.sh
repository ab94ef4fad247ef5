OUT=gpurun_out/r6e; mkdir -p $OUT; export TMPDIR=/tmp
python tools/two_lane_probe.py > $OUT/two_lane_probe.txt 2>&1; cat $OUT/two_lane_probe.txt
for a in "--frames 8 --width 1920 --height 1080" "--frames 32 --width 1920 --height 1080" "--frames 16 --width 576 --height 576" "--frames 8"; do echo "# $a"; python bench.py --steps 10 --no-cpu-baseline --no-e2e --no-refbytes --no-lanes $a 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
