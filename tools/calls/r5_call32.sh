# round 5, call 32 (final tree): functional runs of bench.py --gpus 2 (two ranks on the one GPU of the box over gloo) with this round's bench changes:
# default configuration, with the stock DDP leg, with amp
cd /root/repo
export VBG_DIST_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 4 --warmup 2 --no-amp-leg --no-h2d-leg < /dev/null > gpurun_out/r5c32_2ranks.json 2> gpurun_out/r5c32_2ranks.err; echo "2 ranks default rc=$?"
head -c 1200 gpurun_out/r5c32_2ranks.json; echo; grep -c "" gpurun_out/r5c32_2ranks.json; tail -3 gpurun_out/r5c32_2ranks.err | cut -c1-300
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 4 --warmup 2 --no-amp-leg --no-h2d-leg --stock < /dev/null > gpurun_out/r5c32_2ranks_stock.json 2> gpurun_out/r5c32_2ranks_stock.err; echo "2 ranks + stock DDP leg rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r5c32_2ranks_stock.json')); print(d['value'], d['config'], d.get('stock_loop'))" 2>&1 | cut -c1-900; tail -3 gpurun_out/r5c32_2ranks_stock.err | cut -c1-300
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 4 --warmup 2 --amp --no-h2d-leg < /dev/null > gpurun_out/r5c32_2ranks_amp.json 2> gpurun_out/r5c32_2ranks_amp.err; echo "2 ranks amp rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r5c32_2ranks_amp.json')); print(d['value'], d['config'].get('ranks_in_sync'), d['config'].get('syncbn_comm'))" 2>&1 | cut -c1-300
