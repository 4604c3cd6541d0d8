# round 5, call 43: conv weight-gradient stream levels 0 / 1 (conv + BatchNorm nodes) / 2 (+ plain conv nodes), A/B x3 on one box
cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/r5c43_ab.txt
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5c43_ab.txt; }
for i in 1 2 3; do
  run VBG_CONV_WGRAD_STREAM=0
  run VBG_CONV_WGRAD_STREAM=1
  run VBG_CONV_WGRAD_STREAM=2
done
