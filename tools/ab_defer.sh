#!/bin/bash
# Same-box A/B of seam deferral (srcnn_set_seam_deferral): bench.py with --seam-deferral on / off at four plane sizes, then the
# per-rank stripe projection of the 7680x4320 plane both ways.   tools/ab_defer.sh > profiles/rNN/seam_deferral_ab.txt
export TMPDIR=/tmp
python -m pytest tests/test_gpu_deferral.py -x -q -m gpu 2>&1 | tail -6
for sz in "3840 2160" "1920 1080" "7680 4320" "576 576"; do set -- $sz
 for i in 1 2; do
  for d in on off; do
   echo -n "$1x$2 deferral $d: "; python bench.py --no-cpu-baseline --no-e2e --no-refbytes --no-lanes --sustained-s 0 --seam-deferral $d --width $1 --height $2 --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['output_crc32'])"
  done
 done
done
python tools/stripe_projection.py --ns 1,8 --seam-deferral on 2>&1 | tail -25
python tools/stripe_projection.py --ns 8 --seam-deferral off 2>&1 | tail -14
