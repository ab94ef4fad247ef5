mkdir -p gpurun_out/r6a
python -m pytest tests/test_gpu_deferral.py tests/test_gpu_refbytes.py -x -q -m gpu > gpurun_out/r6a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6a/pytest.log
tail -15 gpurun_out/r6a/pytest.log
python __graft_entry__.py smoke > gpurun_out/r6a/smoke.log 2>&1; tail -2 gpurun_out/r6a/smoke.log
python tools/ab_refbytes.py --sizes 3840x2160,1920x1080,7680x4320 --margins 4 > gpurun_out/r6a/ab_local.txt 2>&1
python tools/ab_refbytes.py --sizes 3840x2160,1920x1080 --margins 4 >> gpurun_out/r6a/ab_local.txt 2>&1
cat gpurun_out/r6a/ab_local.txt
