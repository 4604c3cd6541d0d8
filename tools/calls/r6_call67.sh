cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r6c67_ln.txt
for i in 1 2; do
  for v in 4 2 8 1; do
    VBG_LN_WROWS=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null | grep "^{" > /tmp/x.json
    python -c "import json; d=json.load(open('/tmp/x.json')); print('ln_wrows $v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r6c67_ln.txt
  done
done
