# round 4, call 23: grouped weight-gradient launch of the encoder layers on its own stream (VBG_WGRAD_STREAM), A/B on one box
cd /root/repo
for rep in 1 2 3; do for v in 0 1; do VBG_WGRAD_STREAM=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('WGRAD_STREAM=$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_nt']['frac'])"; done; done
VBG_WGRAD_STREAM=1 timeout 900 python -m pytest tests/test_gpu_full_scale.py tests/test_gpu_ddp.py -x -q -m gpu 2>&1 | tail -3
