#!/bin/bash
# one up-front D2H of the integer inputs: stock loop phases, inference latency, tests that run device-only inputs
cd /root/repo
mkdir -p gpurun_out
python tools/stock_loop_profile.py --steps 10 --optim torch > gpurun_out/r5c26_stock.txt 2> gpurun_out/r5c26_stock.err
python tools/infer_latency.py > gpurun_out/r5c26_infer.txt 2> gpurun_out/r5c26_infer.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg > gpurun_out/r5c26_bench.json 2> gpurun_out/r5c26_bench.err
timeout 1500 python -m pytest tests -m gpu -x -q -k "train_loop or e2e or batch or inference or overlay or full_scale" > gpurun_out/r5c26_tests.txt 2>&1
tail -3 gpurun_out/r5c26_tests.txt
cat gpurun_out/r5c26_stock.txt gpurun_out/r5c26_infer.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c26_bench.json'))
print(d['value'], d['ms_per_step'], d['stock_loop'], d['h2d_inclusive'])
PY
