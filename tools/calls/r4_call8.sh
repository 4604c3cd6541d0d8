# round 4, call 8: functional run of bench.py at 2 ranks (two ranks share the ONE GPU of the box over gloo: RCCL cannot put two ranks on
# one device), default one-communicator configuration and the own-communicator A/B; stock DDP test
cd /root/repo
export VBG_DIST_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --no-amp-leg --no-h2d-leg > gpurun_out/r4c8_bench_2ranks_gloo.json 2> gpurun_out/r4c8_bench_2ranks_gloo.err; echo "2 ranks shared rc=$?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 4 --warmup 2 --no-amp-leg --no-h2d-leg --syncbn-comm own > gpurun_out/r4c8_bench_2ranks_gloo_own.json 2> gpurun_out/r4c8_bench_2ranks_gloo_own.err; echo "2 ranks own rc=$?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 4 --warmup 2 --no-amp-leg --no-h2d-leg --no-ddp-overlap > gpurun_out/r4c8_bench_2ranks_gloo_late.json 2> gpurun_out/r4c8_bench_2ranks_gloo_late.err; echo "2 ranks no-overlap rc=$?"
for f in gpurun_out/r4c8_bench_2ranks_gloo*.json; do python -c "import sys,json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['config'])"; done
tail -3 gpurun_out/r4c8_bench_2ranks_gloo.err | cut -c1-300
unset VBG_DIST_BACKEND
python -m pytest tests/test_gpu_ddp.py -x -q -m gpu -s -k stock > gpurun_out/r4c8_ddp.txt 2>&1; echo "ddp rc=$?"; grep -n "losses stock\|running mean\|rel-L2 of the\|passed\|failed" gpurun_out/r4c8_ddp.txt | cut -c1-500
