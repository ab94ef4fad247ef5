mkdir -p gpurun_out/r6final
python -m pytest tests -x -q -m gpu > gpurun_out/r6final/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6final/pytest.log; tail -4 gpurun_out/r6final/pytest.log
python __graft_entry__.py smoke > gpurun_out/r6final/smoke.log 2>&1; tail -1 gpurun_out/r6final/smoke.log
rm -rf gpurun_out/prof_r06
bash tools/profile_round.sh gpurun_out/prof_r06 > gpurun_out/r6final/profile_round.log 2>&1
tail -c 400 gpurun_out/prof_r06/bench_driver_line.json
