# round 4, call 16: pair planes of the attention output from the forward kernel: attention tests, full-scale parity, A/B against the previous build is by kernel stats
cd /root/repo
python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_full_scale.py -x -q -m gpu -k "cfg2e" -s 2>&1 | grep -E "parameter gradients vs|passed|failed" | cut -c1-160
python -m pytest tests/test_gpu_model.py tests/test_gpu_train_loop.py -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
TOPN=80 bash tools/prof_step.sh 2>&1 | grep -E "split_planes|kernel ms|launches/step|attn_kernel<0"
