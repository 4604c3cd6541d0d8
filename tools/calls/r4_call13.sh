# round 4, call 13: 64-filter tiles for the late trunk stages: kernel tests, batch-8 parity, A/B
cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -3
python -m pytest tests/test_gpu_full_scale.py -x -q -m gpu -k "cfg2e8" -s 2>&1 | grep -E "dispatch seen|passed|failed|gradients vs" | cut -c1-400
for v in 0 1 0 1; do VBG_CONV3_BN64=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BN64=$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_us'])"; done
