# round 6, call 5: the ring WAR fix of conv3x3_kernel's pipelined loop (k-tile 0 read vs the DMA of k-tile 4): 240 default-stream runs, kernel tests
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c05
for i in 1 2 3; do timeout 600 python tools/stream_race_check.py --reps 80 --only-default --offenders 2e-5 --trace 2>/dev/null | grep -v "noise floor" ; done > ${R}_race.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_streams.py -x -q -m gpu -k "conv3 or streams" 2>&1 | tail -6 > ${R}_tests.txt
awk '/conv weight gradients on their stream #/{ tot++; if ($0 ~ /cnn: rel-L2 [0-9.]+e-0[0-5]/) n++ } END{print n+0, "outliers /", tot, "runs"}' ${R}_race.txt
grep "worst over" -A2 ${R}_race.txt | grep cnn; cat ${R}_tests.txt
