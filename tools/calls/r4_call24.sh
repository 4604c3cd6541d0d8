# round 4, call 24: one-product (`amp`) forms of the fast kernels -- kernel tests, the amp golden test, amp leg fast vs generic
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "one_product or amp" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "amp" 2>&1 | tail -5
for v in 1 0 1 0; do VBG_AMP_FAST=$v python bench.py --amp --steps 20 --warmup 5 --no-cpu-baseline --no-h2d-leg 2>gpurun_out/call24_amp$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AMP_FAST=$v', d['value'], d['ms_per_step'], d.get('last_loss'), d['config'].get('workload'))"; done
