# round 6, call 26: AdamW of the encoder layers stepped inside backward (device-gated by the clipping rule): test, A/B x 3
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c26
timeout 900 python -m pytest tests/test_gpu_train_loop.py -x -q -m gpu -k "adamw_stepped or state_dict or gradscaler" 2>&1 | tail -4
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass $1 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager${1:+ off}', d['value'], d['ms_per_step'], d['config']['last_loss'])" | tee -a ${R}_ab.txt; }
rm -f ${R}_ab.txt
for i in 1 2 3; do run --no-eager-opt; run ""; done
