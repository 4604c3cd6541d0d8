# round 6, call 57: hipStream priorities per side stream (the heads' stream's chain is the critical path of the backward's first 2.4 ms)
cd /root/repo
mkdir -p gpurun_out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" | tee gpurun_out/r6c57_prio.txt
for i in 1 2 3; do
  for cfg in "0 0" "-1 0" "-1 1" "0 1"; do
    set -- $cfg
    VBG_PRIO_HEADS=$1 VBG_PRIO_CWGRAD=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null | grep "^{" > /tmp/x.json
    python -c "import json; d=json.load(open('/tmp/x.json')); print('heads $1 cwgrad $2', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r6c57_prio.txt
  done
done
