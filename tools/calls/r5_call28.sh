#!/bin/bash
# LayerNorm backward with plain partial rows instead of slot atomics: tests, kernel times per rows-per-wave, step
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r5c28_ln.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "layernorm or ln or bert or embed" > gpurun_out/r5c28_tests.txt 2>&1 < /dev/null
tail -3 gpurun_out/r5c28_tests.txt
for w in 0 1 2 8; do
  if [ $w = 0 ]; then unset VBG_LN_WROWS; else export VBG_LN_WROWS=$w; fi
  rm -rf /tmp/lnprof
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lnprof -o ln -- python tools/ln_bench.py > gpurun_out/r5c28_ln_w$w.txt 2>&1 < /dev/null
  f=$(find /tmp/lnprof -name "*kernel_stats.csv" | head -1)
  echo "== wrows $w" >> gpurun_out/r5c28_ln.txt
  if [ -n "$f" ]; then grep -E "dropout_add_ln|ln_fold" "$f" | awk -F'","' '{print $1, $2, $4}' | cut -c1-150 >> gpurun_out/r5c28_ln.txt; fi
done
unset VBG_LN_WROWS
cat gpurun_out/r5c28_ln.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-stock-leg --no-h2d-leg > gpurun_out/r5c28_bench.json 2> gpurun_out/r5c28_bench.err < /dev/null
python -c "
import json; d=json.load(open('gpurun_out/r5c28_bench.json')); print(d['value'], d['ms_per_step'])"
