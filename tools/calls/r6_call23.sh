# round 6, call 23: when the host enqueues the backward of the CNN's first stage (autograd sequence numbers): after the encoder's top n layers
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c23
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config']['last_loss'])" | tee -a ${R}_ab.txt; }
rm -f ${R}_ab.txt
for i in 1 2; do for n in -1 0 2 4 6 8 10; do run VBG_STAGE1_BWD_AFTER=$n; done; done
