cd /root/repo
timeout 900 python -m pytest "tests/test_gpu_full_scale.py::test_full_scale_amp_one_product_forms" -q -m gpu --tb=short -rf -s 2>&1 | grep "amp:\|passed\|failed\|assert\|Error" | cut -c1-500
