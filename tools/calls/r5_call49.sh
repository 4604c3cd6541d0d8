# round 5, call 49: sanity of the rebuilt library at HEAD: smoke, a fast subset of the GPU suite, a short bench
cd /root/repo
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -1
timeout 900 python -m pytest tests -m gpu -x -q -k "conv3 or layernorm or ln or train_loop or side_streams" 2>&1 < /dev/null | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
