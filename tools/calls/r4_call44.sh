# round 4, call 44: attention workgroups of 64 own rows (two waves) against 128 (four): tests, kernels stand-alone, step A/B
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -3
for w in 128 64 128 64; do echo "== wg_rows $w"; WG_ROWS=$w python tools/attn_bench.py 2>/dev/null | grep dropout; done
for rep in 1 2 3; do for w in 128 64; do VBG_ATTN_WG_ROWS=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('WG_ROWS=$w', d['value'], d['ms_per_step'])"; done; done
