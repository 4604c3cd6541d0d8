# round 6, call 28: flakiness check -- the -m gpu suite twice more, the default bench line as the driver runs it
cd /root/repo
mkdir -p gpurun_out
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -1; done | tee gpurun_out/r6c28_pytest.txt
( time python bench.py > gpurun_out/r6c28_bench.json 2> gpurun_out/r6c28_bench.err ) 2>&1 | grep real
python -c "import json; d=json.load(open('gpurun_out/r6c28_bench.json')); print(d['value'], d['ms_per_step'], d['stock_loop']['value'], d['amp']['value'], d['roofline']['frac'], d['roofline'].get('frac_trace'), d['cpu_baseline']['value'])"
python __graft_entry__.py smoke 2>&1 | tail -1
