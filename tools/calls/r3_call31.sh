set -x
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3 or conv2d" 2>&1 | tail -15 > gpurun_out/call31_pytest.txt
cat gpurun_out/call31_pytest.txt
python tools/conv3_bench.py > gpurun_out/call31_conv3_shapes.txt 2>&1
tail -30 gpurun_out/call31_conv3_shapes.txt
bash tools/run_ab.sh VBG_CONV3_SPLITK
