# round 6, call 25: the CNN's first stage enqueued BETWEEN encoder layers in the forward (VBG_STAGE1_FWD_AFTER=k), A/B x 2
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c25
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config']['last_loss'])" | tee -a ${R}_ab.txt; }
rm -f ${R}_ab.txt
for i in 1 2; do for n in -1 0 1 2 4 6; do run VBG_STAGE1_FWD_AFTER=$n; done; done
