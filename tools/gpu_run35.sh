#!/bin/bash
python -m pytest tests/test_gpu_pipeline.py tests/test_cli.py -m gpu -x -q 2>&1 | tail -3
kms() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['output_crc32'][0])"; }
for i in 1 2 3; do
echo -n "pipeline fused launches: "; kms --path pipeline --steps 50
echo -n "pipeline 3 kernels:      "; SRCNN_DEBUG_PIPE3=1 kms --path pipeline --steps 50
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_pipe2 -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --path pipeline --steps 20 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; find gpurun_out/trace_pipe2 -name "*kernel_stats.csv" | head -1 | xargs cat | cut -d, -f1-4 | head -8
