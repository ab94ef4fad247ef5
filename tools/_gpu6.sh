OUT=gpurun_out/r6f; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_deferral.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
python tools/two_lane_probe.py 7680x540,2560x1440,960x540 > $OUT/two_lane_probe2.txt 2>&1; cat $OUT/two_lane_probe2.txt
for a in "--width 576 --height 576" "--width 1920 --height 1080" ""; do echo "# $a"; python bench.py --steps 20 --no-cpu-baseline --no-e2e --no-refbytes $a 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('two_lanes'))"; done
