# round 6, call 29: the loss tail as one selected-CE node per loss (instead of one per category): parity tests, then A/B on one box
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "loss or golden or full_scale or e2e or train" 2>&1 | tail -2 | tee gpurun_out/r6c29_pytest.txt
: > gpurun_out/r6c29_ab.txt
for i in 1 2 3; do
  for v in 0 1; do
    VBG_LOSS_FUSE=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-single-stream-pass > /tmp/b.json 2>/dev/null
    python -c "import json; d=json.load(open('/tmp/b.json')); print('fuse$v', d['value'], d['ms_per_step'], d['stock_loop']['value'], d.get('host'))" | tee -a gpurun_out/r6c29_ab.txt
  done
done
