cd /root/repo
timeout 900 python -m pytest tests/test_gpu_attention.py -q -m gpu --tb=line -rf -s -k "pair_form or one_product" 2>&1 | grep "pair form\|one-product\|passed\|failed" | cut -c1-300
