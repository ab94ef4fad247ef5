#!/bin/bash
# Host cost of ONE row-striped step (configs[3]: a 7680x4320 plane) on the 1-GPU box, for 2 / 4 / 8 ranks or contexts that
# SHARE the GPU: the kernels then run one after the other, so `host_us_per_step` -- the time to queue a step: launches, halo
# posts, band copies -- is what an N-GPU node would pay per step next to 3.77 / N ms of kernel per rank.
#   tools/stripe_overhead.sh [OUT]      (default gpurun_out/stripe_overhead.txt)
OUT=${1:-gpurun_out/stripe_overhead.txt}
mkdir -p $(dirname $OUT)
pick() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'n', d['n_gpus'], 'host_us_per_step', d.get('host_us_per_step'), 'ms_per_step', d['ms_per_step'], 'rank0_kernel_ms', d['roofline']['kernel_ms'],
      'transport', d['distributed']['halo_transport'] if 'distributed' in d else '-', 'crc', d['config']['output_crc32'])"; }
{
echo "# 7680x4320 plane, ranks / contexts share ONE MI355X; steps 30, warmup 5"
python bench.py --no-cpu-baseline --no-e2e --workload stripe --width 7680 --height 4320 --steps 30 | pick "py-1rank"
for n in 2 4 8; do
  python bench.py --no-cpu-baseline --gpus $n --shared-gpu --backend gloo --workload stripe --width 7680 --height 4320 --steps 30 | pick "py-ranks(gloo-staged halo)"
  python bench.py --no-cpu-baseline --gpus $n --shared-gpu --host cxx --workload stripe --width 7680 --height 4320 --steps 30 | pick "cxx-contexts(hipMemcpy halo)"
  # 2 streams per context: beyond the runtime's 4 hardware queues per device the streams of ONE device share queues and a
  # cross-stream event wait can block the host -- an artefact of sharing one GPU (a node has 2 streams per device)
  GPU_MAX_HW_QUEUES=32 python bench.py --no-cpu-baseline --gpus $n --shared-gpu --host cxx --workload stripe --width 7680 --height 4320 --steps 30 | pick "cxx-contexts(hipMemcpy halo),GPU_MAX_HW_QUEUES=32"
done
for n in 2 8; do
  python bench.py --no-cpu-baseline --gpus $n --shared-gpu --host cxx --workload frames --steps 30 | pick "cxx-frames-3840x2160"
done
} > $OUT 2>&1
cat $OUT
