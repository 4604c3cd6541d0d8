R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_full_scale.py -m gpu -x -q -k "ln or cfg2e or cfg5e" 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
