# round 4, call 42: q / k / v bias gradients from the attention backward kernels (VBG_ATTN_COLSUM) -- tests, step A/B
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_full_scale.py -x -q -m gpu -k "cfg2e8 or cfg2e or amp" 2>&1 | tail -3
for rep in 1 2 3; do for v in 0 1; do VBG_ATTN_COLSUM=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ATTN_COLSUM=$v', d['value'], d['ms_per_step'])"; done; done
