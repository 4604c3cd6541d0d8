# round 5, call 39: conv weight gradients on a stream of their own beside the input gradient (VBG_CONV_WGRAD_STREAM), and the encoder's grouped weight gradients likewise (VBG_WGRAD_STREAM): A/B x2, tests with both on
cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/r5c39_ab.txt
run() { env $1 $2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5c39_ab.txt; }
for i in 1 2; do
  run VBG_NOP=1 VBG_NOP2=1
  run VBG_CONV_WGRAD_STREAM=1 VBG_NOP2=1
  run VBG_WGRAD_STREAM=1 VBG_NOP2=1
  run VBG_CONV_WGRAD_STREAM=1 VBG_WGRAD_STREAM=1
done
run VBG_NOP=1 VBG_NOP2=1
VBG_CONV_WGRAD_STREAM=1 VBG_WGRAD_STREAM=1 timeout 1500 python -m pytest tests -m gpu -x -q -k "full_scale or train_loop or ddp or side_streams or e2e" > gpurun_out/r5c39_tests.txt 2>&1 < /dev/null
tail -2 gpurun_out/r5c39_tests.txt
