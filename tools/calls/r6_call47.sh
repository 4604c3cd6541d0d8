cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r6c47_pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-single-stream-pass > /tmp/b.json 2>/dev/null
python -c "import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['ms_per_step'], d['stock_loop']['value'], d.get('host'))" | tee gpurun_out/r6c47_bench.txt
