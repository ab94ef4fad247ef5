OUT=gpurun_out/r6d; mkdir -p $OUT; export TMPDIR=/tmp
python tests/checks/adversarial_gpu_ratio.py 200 100 2 > $OUT/adversarial_gpu_ratio.txt 2>&1; grep -v "step " $OUT/adversarial_gpu_ratio.txt | tail -32
