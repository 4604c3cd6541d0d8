cd /root/repo
timeout 900 python -m pytest tests/test_gpu_attention.py -q -m gpu --tb=short -rf -s 2>&1 | grep "pair form: O\|one-product\|passed\|failed\|Error\|assert\|FAILED" | cut -c1-300
python tools/attn_bench.py 2>&1 | grep "^form"
timeout 1200 python -m pytest "tests/test_gpu_full_scale.py::test_full_scale_every_gradient_vs_reference[cfg2e8]" -q -m gpu --tb=short -rf -s 2>&1 | grep "parameter gradients\|passed\|failed" | cut -c1-200
bash tools/run_ab.sh VBG_ATTN_PAIR 2>&1 | grep "VBG_ATTN_PAIR="
