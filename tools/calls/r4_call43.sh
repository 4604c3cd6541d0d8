# round 4, call 43: final check of the committed tree -- whole GPU suite, smoke, the default bench line
cd /root/repo
timeout 2700 python -m pytest tests -q -m gpu --tb=short -rf 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err ) 2>&1 | grep real
python -c "import json; d=json.load(open('gpurun_out/final_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_nt']['frac'], d['amp']['value'], d['cpu_baseline']['value'])"
