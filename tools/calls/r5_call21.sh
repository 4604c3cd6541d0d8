# round 5, call 21: wall time of the default bench run (what the driver executes), RCCL test after the fallback edit
cd /root/repo
( time python bench.py > gpurun_out/r5c21_bench.json 2> gpurun_out/r5c21_bench.err ) 2>&1 | grep real
python -c "import json; d=json.load(open('gpurun_out/r5c21_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stock_loop']['value'], d['amp']['value'], d['cpu_baseline']['value'], d['cpu_baseline']['sweep'])"
timeout 600 python -m pytest tests/test_gpu_ddp.py -q -m gpu --tb=short -k rccl 2>&1 | tail -2
