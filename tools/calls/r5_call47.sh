# round 5, call 47: the final tree once more -- full GPU suite, smoke, default bench, and the functional two-rank bench runs (gloo, one GPU)
cd /root/repo
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r5c47_tests.txt 2>&1 < /dev/null
tail -2 gpurun_out/r5c47_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5c47_smoke.txt 2>&1 < /dev/null
tail -1 gpurun_out/r5c47_smoke.txt
( time timeout 900 python bench.py ) > gpurun_out/r5c47_bench.json 2> gpurun_out/r5c47_bench.err < /dev/null
tail -4 gpurun_out/r5c47_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r5c47_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'stock', d['stock_loop']['value'], 'amp', d['amp']['value'], 'h2d', d['h2d_inclusive']['value'], d['roofline']['frac'], d['roofline'].get('single_stream'))"
export VBG_DIST_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 4 --warmup 2 --no-amp-leg --no-h2d-leg < /dev/null > gpurun_out/r5c47_2ranks.json 2> gpurun_out/r5c47_2ranks.err; echo "2 ranks default rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r5c47_2ranks.json')); print(d['value'], d['config'].get('ranks_in_sync'), d['config'].get('syncbn_comm'))" 2>&1 | cut -c1-300
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 2 --steps 4 --warmup 2 --no-amp-leg --no-h2d-leg --stock < /dev/null > gpurun_out/r5c47_2ranks_stock.json 2> gpurun_out/r5c47_2ranks_stock.err; echo "2 ranks + stock DDP leg rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r5c47_2ranks_stock.json')); print(d['value'], d['config'].get('ranks_in_sync'), d.get('stock_loop'))" 2>&1 | cut -c1-600
