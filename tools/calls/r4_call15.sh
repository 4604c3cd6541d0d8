# round 4, call 15: LayerNorm backward -> bound-scaled pair planes: kernel test, full-scale parity, training-loop tests, A/B
cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm_backward_bound or bound_scaled or pair_form" 2>&1 | tail -4
python -m pytest tests/test_gpu_full_scale.py -x -q -m gpu -k "cfg2e or cfg4e" -s > gpurun_out/r4c15_full.txt 2>&1; echo "full-scale rc=$?"; grep -n "parameter gradients vs\|passed\|failed" gpurun_out/r4c15_full.txt | cut -c1-200
python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3
for v in 0 1 0 1; do VBG_BOUND_PLANES=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BOUND_PLANES=$v', d['value'], d['ms_per_step'])"; done
