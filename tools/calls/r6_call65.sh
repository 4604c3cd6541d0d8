cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "slab" 2>&1 | grep -E "^E |passed|failed" | head
