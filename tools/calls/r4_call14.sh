# round 4, call 14: weight-gradient kernel with the loads two k-tiles ahead: tests, shapes, A/B
cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3_wgrad or conv3x3_dispatch" 2>&1 | tail -3
VBG_CONV3W_DEEP=0 python tools/conv3_bench.py 2>&1 | grep "f16x2, slabs" > gpurun_out/r4c14_w0.txt
VBG_CONV3W_DEEP=1 python tools/conv3_bench.py 2>&1 | grep "f16x2, slabs" > gpurun_out/r4c14_w1.txt
paste -d'|' <(cut -c1-95 gpurun_out/r4c14_w0.txt) <(cut -c60-95 gpurun_out/r4c14_w1.txt)
for v in 0 1 0 1; do VBG_CONV3W_DEEP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('W_DEEP=$v', d['value'], d['ms_per_step'])"; done
