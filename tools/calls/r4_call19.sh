# round 4, call 19: reduction order of the row-reuse forward kernels, A/B on one box
cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -2
VBG_CONV3_KORDER=0 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -2
for v in 0 1 0 1; do VBG_CONV3_KORDER=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('KORDER=$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_us'])"; done
