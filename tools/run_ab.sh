# A/B of one environment switch on the headline step: tools/run_ab.sh VAR  (runs VAR=0, VAR=1, VAR=0, VAR=1)
set -x
V=$1
mkdir -p gpurun_out/ab
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg"
for i in 1 2; do for x in 0 1; do env $V=$x $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$x', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['last_loss'])" | tee -a gpurun_out/ab/$V.txt; done; done
