OUT=gpurun_out/r6c; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_refbytes.py tests/test_gpu_split16.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
python tools/ab_refbytes.py --sizes 3840x2160,1920x1080 --margins 4 --modes refbytes,refbytes16 > $OUT/ab_local.txt 2>&1; cat $OUT/ab_local.txt
python tests/checks/adversarial_gpu_ratio.py 150 100 1 > $OUT/adversarial_gpu_ratio.txt 2>&1; grep -v "step " $OUT/adversarial_gpu_ratio.txt | tail -25
for sz in "3840 2160" "1920 1080" "1280 720" "576 576" "7680 540"; do python tools/diag_light.py $sz > $OUT/diag_light_$(echo $sz | tr ' ' x).txt 2>&1; done
head -8 $OUT/diag_light_576x576.txt $OUT/diag_light_1920x1080.txt $OUT/diag_light_7680x540.txt
