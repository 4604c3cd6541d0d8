cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_train_loop.py -q -m gpu --tb=short -rf -s 2>&1 | grep "stock loop\|worst \|passed\|failed\|FAILED\|assert \|Error" | cut -c1-400
