#!/bin/bash
# LayerNorm backward kernel times: stand-alone per rows-per-wave, and inside the step
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r5c29_ln.txt
for w in 0 1 2 8; do
  if [ $w = 0 ]; then unset VBG_LN_WROWS; else export VBG_LN_WROWS=$w; fi
  rm -rf /tmp/lnprof
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lnprof -o ln -- python tools/ln_bench.py > /dev/null 2>&1 < /dev/null
  echo "== wrows $w" >> gpurun_out/r5c29_ln.txt
  python tools/kstat.py /tmp/lnprof "dropout_add_ln|ln_fold" >> gpurun_out/r5c29_ln.txt
done
unset VBG_LN_WROWS
cat gpurun_out/r5c29_ln.txt
rm -rf /tmp/stepprof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stepprof -o e -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg > /dev/null 2>&1 < /dev/null
python tools/kstat.py /tmp/stepprof "dropout_add_ln|ln_fold|embed_ln" > gpurun_out/r5c29_step.txt
cat gpurun_out/r5c29_step.txt
