# round 5, call 46: conv weight-gradient stream for every node vs only for nodes below a pixel count, at cfg2 and cfg5
cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/r5c46_ab.txt
run() { env $2 $3 timeout 400 python bench.py --shape $1 --steps $4 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 $3', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5c46_ab.txt; }
for i in 1 2; do
  run cfg2 VBG_CONV_WGRAD_STREAM=2 VBG_CONV_WGRAD_STREAM_MAXPIX=0 20
  run cfg2 VBG_CONV_WGRAD_STREAM=2 VBG_CONV_WGRAD_STREAM_MAXPIX=32768 20
  run cfg2 VBG_CONV_WGRAD_STREAM=2 VBG_CONV_WGRAD_STREAM_MAXPIX=8192 20
  run cfg5 VBG_CONV_WGRAD_STREAM=0 VBG_OVERLAP=0 6
  run cfg5 VBG_CONV_WGRAD_STREAM=0 VBG_OVERLAP=1 6
  run cfg5 VBG_CONV_WGRAD_STREAM=2 VBG_CONV_WGRAD_STREAM_MAXPIX=0 6
  run cfg5 VBG_CONV_WGRAD_STREAM=2 VBG_CONV_WGRAD_STREAM_MAXPIX=65536 6
  run cfg5 VBG_CONV_WGRAD_STREAM=2 VBG_CONV_WGRAD_STREAM_MAXPIX=32768 6
done
