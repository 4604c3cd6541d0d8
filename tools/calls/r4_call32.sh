# round 4, call 32: 64-deep k-tiles in the step -- test, A/B (VBG_PAIR_BK64), amp
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "deep_ktiles or pair_form or one_product or bound_scaled" 2>&1 | tail -5
for rep in 1 2 3; do for v in 0 1; do VBG_PAIR_BK64=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BK64=$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_nt']['frac'])"; done; done
for v in 0 1; do VBG_PAIR_BK64=$v python bench.py --amp --steps 20 --warmup 5 --no-cpu-baseline --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('amp BK64=$v', d['value'], d['ms_per_step'])"; done
