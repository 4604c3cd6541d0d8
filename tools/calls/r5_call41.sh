# round 5, call 41: conv weight gradients on their own stream by default: full GPU suite, smoke, default bench
cd /root/repo
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r5c41_tests.txt 2>&1 < /dev/null
tail -2 gpurun_out/r5c41_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5c41_smoke.txt 2>&1 < /dev/null
tail -1 gpurun_out/r5c41_smoke.txt
( time timeout 900 python bench.py ) > gpurun_out/r5c41_bench.json 2> gpurun_out/r5c41_bench.err < /dev/null
tail -4 gpurun_out/r5c41_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c41_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'stock', d['stock_loop']['value'], 'amp', d['amp']['value'], 'h2d', d['h2d_inclusive']['value'])
for k in ('roofline','roofline_conv3','roofline_nt'):
    if k in d: print(k, d[k]['frac'], d[k]['avg_us'], d[k].get('single_stream'))
PY
