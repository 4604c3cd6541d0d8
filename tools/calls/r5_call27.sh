#!/bin/bash
# full GPU suite + smoke + default bench on the final tree
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5c27_tests.txt 2>&1
tail -3 gpurun_out/r5c27_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5c27_smoke.txt 2>&1
tail -1 gpurun_out/r5c27_smoke.txt
( time python bench.py ) > gpurun_out/r5c27_bench.json 2> gpurun_out/r5c27_bench.err
tail -4 gpurun_out/r5c27_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c27_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['stock_loop']['value'], d['amp']['value'] if isinstance(d.get('amp'),dict) else d.get('amp'))
PY
