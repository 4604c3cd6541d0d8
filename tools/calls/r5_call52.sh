# round 5, call 52: more switches whose verdicts predate the stream defaults, A/B x2 on one box
cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/r5c52_ab.txt
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5c52_ab.txt; }
for i in 1 2; do
  run VBG_NOP=1
  run VBG_STREAMK=1
  run VBG_CONV3_BN64=0
  run VBG_CONV3_SPLITK=0
  run VBG_PAIR_DEEP=0
  run VBG_CONV3_MIN_TILES_FWD=128
done
run VBG_NOP=1
