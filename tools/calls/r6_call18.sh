# round 6, call 18: with five streams in a step -- hardware queues (GPU_MAX_HW_QUEUES), the encoder's weight-gradient stream again: A/B x 2
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c18
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a ${R}_ab.txt; }
rm -f ${R}_ab.txt
for i in 1 2; do run VBG_NOP=1; run GPU_MAX_HW_QUEUES=8; run GPU_MAX_HW_QUEUES=2; run VBG_WGRAD_STREAM=1; run VBG_CONV_WGRAD_STREAM=1; done
run VBG_NOP=1
