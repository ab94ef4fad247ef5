mkdir -p gpurun_out/r6k
python -m pytest tests -x -q -m gpu > gpurun_out/r6k/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6k/pytest.log; tail -6 gpurun_out/r6k/pytest.log
bash tools/profile_round.sh gpurun_out/prof_r06 > gpurun_out/r6k/profile_round.log 2>&1
tail -5 gpurun_out/prof_r06/pmc_summarize.log
cat gpurun_out/prof_r06/bench_driver_line.json | tail -c 600
