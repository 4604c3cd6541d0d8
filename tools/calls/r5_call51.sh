# round 5, call 51: the encoder's stream as a high-priority queue? A/B x3
cd /root/repo
mkdir -p gpurun_out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5c51_ab.txt; }
rm -f gpurun_out/r5c51_ab.txt
for i in 1 2 3; do
  run VBG_SIDE_PRIORITY=0
  run VBG_SIDE_PRIORITY=-1
done
