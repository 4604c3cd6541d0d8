# round 6, call 17: transposed weight planes / L1 bounds refreshed behind the encoder's forward instead of in front of its backward: A/B x 3
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c17
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config']['last_loss'])" | tee -a ${R}_ab.txt; }
rm -f ${R}_ab.txt
for i in 1 2 3; do run VBG_PREFETCH_BWD=0; run VBG_PREFETCH_BWD=1; done
timeout 900 python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_streams.py -x -q -m gpu 2>&1 | tail -3
