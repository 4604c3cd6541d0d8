# round 6, call 58: the segmentation head's backward gated behind the small front part of the RoI branch's backward
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r6c58_gate.txt
for i in 1 2 3; do
  for v in 0 1; do
    VBG_CLS_GATE=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null | grep "^{" > /tmp/x.json
    python -c "import json; d=json.load(open('/tmp/x.json')); print('gate $v', d['value'], d['ms_per_step'], d['config'].get('last_loss'))" | tee -a gpurun_out/r6c58_gate.txt
  done
done
