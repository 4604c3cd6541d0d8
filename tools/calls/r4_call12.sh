# round 4, call 12: faster column-L1 kernel; A/B of the bound-scaled dL/dh planes again
cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bound_scaled" 2>&1 | tail -2
for v in 0 1 0 1; do VBG_BOUND_PLANES=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BOUND_PLANES=$v', d['value'], d['ms_per_step'])"; done
TOPN=70 bash tools/prof_step.sh 2>&1 | grep -E "col_l1|kernel ms|split_planes_pair"
