# round 5, call 15: fused attention on two fp16 pieces (FORM 1) and on the hi pieces alone (FORM 2): kernel tests, timing, model tests, A/B
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_attention.py -q -m gpu --tb=short -rf -s 2>&1 | grep "pair form\|one-product\|passed\|failed\|Error\|assert\|FAILED" | cut -c1-300
python tools/attn_bench.py 2>&1 | grep "^form"
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -rf -x 2>&1 | tail -4
timeout 1200 python -m pytest "tests/test_gpu_full_scale.py::test_full_scale_every_gradient_vs_reference[cfg2e8]" "tests/test_gpu_full_scale.py::test_full_scale_amp_one_product_forms" "tests/test_gpu_full_scale.py::test_full_scale_every_gradient_vs_reference[cfg2e]" -q -m gpu --tb=short -rf -s 2>&1 | grep "parameter gradients\|dispatch seen\|cosines\|train loss\|passed\|failed\|Error\|assert" | cut -c1-700
bash tools/run_ab.sh VBG_ATTN_PAIR 2>&1 | grep "VBG_ATTN_PAIR="
