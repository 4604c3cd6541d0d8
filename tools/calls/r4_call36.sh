# round 4, call 36: functional 2-rank runs of bench.py on the final build (two ranks share the one GPU over gloo): default configuration,
# with the amp leg, and the headline as amp
cd /root/repo
export VBG_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 4 --warmup 2 --no-h2d-leg > gpurun_out/r4c36_2ranks.json 2> gpurun_out/r4c36_2ranks.err; echo "2 ranks default rc=$?"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 2 --steps 4 --warmup 2 --no-h2d-leg --amp > gpurun_out/r4c36_2ranks_amp.json 2> gpurun_out/r4c36_2ranks_amp.err; echo "2 ranks amp rc=$?"
for f in gpurun_out/r4c36_2ranks.json gpurun_out/r4c36_2ranks_amp.json; do python -c "import sys,json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d.get('amp'), d['config'])"; done
tail -3 gpurun_out/r4c36_2ranks.err | cut -c1-300
