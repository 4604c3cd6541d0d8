# round 4, call 11: bound-scaled pair planes of dL/dh from the GEMM epilogue: kernel test, full-scale parity with it on, A/B in the step
cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bound_scaled or pair_form or roi_align" -s 2>&1 | tail -6
python -m pytest tests/test_gpu_full_scale.py -x -q -m gpu -k "cfg2e or cfg3e" -s > gpurun_out/r4c11_full.txt 2>&1; echo "full-scale rc=$?"; grep -n "parameter gradients vs\|passed\|failed" gpurun_out/r4c11_full.txt | cut -c1-330
for v in 0 1 0 1; do VBG_BOUND_PLANES=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BOUND_PLANES=$v', d['value'], d['ms_per_step'])"; done
