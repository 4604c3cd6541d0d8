# round 5, call 1: model-owned flat storage -- the new tests, the loops that touch it, then the stock_loop leg of the bench
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_ddp.py "tests/test_gpu_full_scale.py::test_stock_loop_takes_the_all_pair_backward" "tests/test_gpu_full_scale.py::test_full_scale_every_gradient_vs_reference[cfg2e8]" -q -m gpu --tb=short -rf -x -s 2>&1 | grep -v "^$" | tail -40
( time python bench.py --no-cpu-baseline --no-amp-leg > gpurun_out/r5c1_bench.json 2> gpurun_out/r5c1_bench.err ) 2>&1 | grep real
tail -5 gpurun_out/r5c1_bench.err
python -c "import json; d=json.load(open('gpurun_out/r5c1_bench.json')); print(d['value'], d['ms_per_step'], d.get('stock_loop'), d.get('h2d_inclusive'))"
VBG_HOME=0 python bench.py --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VBG_HOME=0', d['value'], d.get('stock_loop'))"
