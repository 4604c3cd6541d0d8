set -x
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3 or conv2d" 2>&1 | tail -5 > gpurun_out/call32_pytest.txt
cat gpurun_out/call32_pytest.txt
python -m pytest tests/test_gpu_full_scale.py tests/test_gpu_e2e.py -q -x 2>&1 | tail -5 > gpurun_out/call32_pytest2.txt
cat gpurun_out/call32_pytest2.txt
bash tools/run_ab.sh VBG_CONV3_SPLITK
