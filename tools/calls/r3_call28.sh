R=$GRAFT_REPO_ROOT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
VBG_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 --no-amp-leg --no-h2d-leg 2> gpurun_out/call28.err | python -c "
import sys,json
s=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]
d=json.loads(s[-1]); print(d['value'], d['ms_per_step'], d['config'])"
tail -3 gpurun_out/call28.err
