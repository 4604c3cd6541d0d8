# round 6, call 52: the headline step of the tree at 493100c (before the layer entry / inference work) against the current tree, same box
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r6c52_ab.txt
for i in 1 2 3; do
  for t in ab_old .; do
    ( cd $t && python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null | grep "^{" > /tmp/x.json )
    python -c "import json; d=json.load(open('/tmp/x.json')); print('$t', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r6c52_ab.txt
  done
done
