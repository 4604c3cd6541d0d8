# round 6, call 19: GPU_MAX_HW_QUEUES sweep (default 4; 8 was 28 % SLOWER, 2 was 3.5 % slower)
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c19
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a ${R}_ab.txt; }
rm -f ${R}_ab.txt
for i in 1 2; do run VBG_NOP=1; run GPU_MAX_HW_QUEUES=3; run GPU_MAX_HW_QUEUES=4; run GPU_MAX_HW_QUEUES=5; run GPU_MAX_HW_QUEUES=6; done
