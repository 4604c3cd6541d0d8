cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fp16_pair_form_of_the_forward" 2>&1 | grep -E "^E |assert|Error|passed|failed" | head -20
