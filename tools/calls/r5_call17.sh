cd /root/repo
for d in 0 16 32 48; do echo "debug $d"; VBG_ATTN_DEBUG=$d timeout 900 python -m pytest tests/test_gpu_attention.py -q -m gpu --tb=line -rf -s -k "pair_form_vs" 2>&1 | grep "pair form" | cut -c1-200 | head -2; done
