cd /root/repo
timeout 900 python -m pytest tests/test_gpu_layer_entry.py -x -q 2>&1 | grep -v Warning | tail -15
