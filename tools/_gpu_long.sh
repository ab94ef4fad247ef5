OUT=gpurun_out/r6long; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_hardening.py -x -q -m gpu > $OUT/pytest_hardening.log 2>&1; tail -3 $OUT/pytest_hardening.log
python tests/checks/adversarial_gpu_ratio.py 480 100 3 > $OUT/adversarial_gpu_ratio_long.txt 2>&1; grep -v "step " $OUT/adversarial_gpu_ratio_long.txt | tail -12
python tests/checks/soak_models.py 600 211 > $OUT/soak_models_long.txt 2>&1; tail -2 $OUT/soak_models_long.txt
python tests/checks/soak.py 400 105 > $OUT/soak_long.txt 2>&1; tail -3 $OUT/soak_long.txt
