# round 5, call 36: switches whose verdicts predate this round's kernels, A/B on one box (two alternations)
cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/r5c36_ab.txt
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5c36_ab.txt; }
for i in 1 2; do
  run VBG_NOP=1
  run VBG_OVERLAP=1
  run VBG_WGRAD_STREAM=1
  run VBG_CONV3W_BLOCKS=512
  run VBG_CONV3W_REDUCE_PAR=8
  run VBG_CONV3W_REDUCE_PAR=16
  run VBG_CONV3W_MIN=4
done
run VBG_NOP=1
