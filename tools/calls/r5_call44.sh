# round 5, call 44: is the bimodal step time with three compute streams a hardware-queue assignment? GPU_MAX_HW_QUEUES 4 (default) vs 8, six alternations
cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/r5c44_ab.txt
run() { env $1 $2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5c44_ab.txt; }
for i in 1 2 3 4 5 6; do
  run VBG_CONV_WGRAD_STREAM=2 VBG_NOP=1
  run VBG_CONV_WGRAD_STREAM=2 GPU_MAX_HW_QUEUES=8
  run VBG_CONV_WGRAD_STREAM=0 VBG_NOP=1
done
