cd /root/repo; mkdir -p gpurun_out
timeout 600 python tools/stream_race_check.py --reps 40 --only-default --offenders 2e-5 2>/dev/null | grep -v "noise floor #" | tail -4
timeout 1500 python -m pytest tests/test_gpu_streams.py tests/test_gpu_model.py tests/test_gpu_train_loop.py tests/test_gpu_full_scale.py -x -q -m gpu -k "streams or side or stock or cfg2e8 or train_loop or inference" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
