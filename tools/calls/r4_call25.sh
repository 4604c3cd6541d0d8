# round 4, call 25: one-product forms -- kernel tests again, then the whole GPU suite
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "one_product or amp" 2>&1 | tail -15
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
