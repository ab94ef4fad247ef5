#!/bin/bash
# first GPU pass of the round: tests, bench lines of the new paths, PMC of the unfused kernels
OUT=gpurun_out/r02a
ROOT=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_configs.py > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --gpus 2 --shared-gpu --backend gloo > $OUT/bench_2rank_frames.json 2> $OUT/bench_2rank_frames.err
python bench.py --gpus 2 --shared-gpu --backend gloo --workload stripe --width 7680 --height 4320 > $OUT/bench_2rank_stripe.json 2> $OUT/bench_2rank_stripe.err
python bench.py --gpus 2 --shared-gpu --backend gloo --workload stripe --width 7680 --height 4320 --no-overlap > $OUT/bench_2rank_stripe_noov.json 2>> $OUT/bench_2rank_stripe.err
python bench.py --path surface --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_surface.json 2> $OUT/bench_surface.err
python bench.py --path host --steps 10 --no-cpu-baseline > $OUT/bench_host.json 2>&1
g++ -std=c++17 -pthread -Iinclude tools/host_demo_multi.cpp -Lsrcnn_cpp_amd -lsrcnn_amd -Wl,-rpath,$ROOT/srcnn_cpp_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/host_demo_multi \
  && /tmp/host_demo_multi srcnn_cpp_amd/data/srcnn915_weights.f32 3840 2160 8 /tmp/o.u8 0 0 > $OUT/host_demo_multi.txt 2>&1
for grp in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/pmc_unfused_$grp -o pmc -- \
      python3 $ROOT/bench.py --path unfused --steps 3 --warmup 1 --no-cpu-baseline ) > $OUT/pmc_unfused_$grp.log 2>&1
done
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_unfused -o trace -- \
      python3 $ROOT/bench.py --path unfused --steps 5 --warmup 1 --no-cpu-baseline ) > $OUT/trace_unfused.log 2>&1
python - <<'PY'
import csv, glob, collections
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r02a/pmc_unfused_{grp}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(grp, k, len(v), sum(v) / len(v))
PY
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
cat $OUT/bench_*.json | cut -c1-600
cat $OUT/host_demo_multi.txt
