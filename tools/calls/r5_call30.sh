#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "layernorm or ln or bert or embed" > gpurun_out/r5c30_tests.txt 2>&1 < /dev/null
tail -2 gpurun_out/r5c30_tests.txt
rm -rf /tmp/lnprof
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lnprof -o ln -- python tools/ln_bench.py > /dev/null 2>&1 < /dev/null
python tools/kstat.py /tmp/lnprof "dropout_add_ln|ln_fold" > gpurun_out/r5c30_ln.txt
cat gpurun_out/r5c30_ln.txt
rm -rf /tmp/stepprof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stepprof -o e -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg > /dev/null 2>&1 < /dev/null
python tools/kstat.py /tmp/stepprof "dropout_add_ln|ln_fold|embed_ln" > gpurun_out/r5c30_step.txt
cat gpurun_out/r5c30_step.txt
