cd /root/repo; mkdir -p gpurun_out
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host'])"; done | tee gpurun_out/r6c11_host.txt
timeout 300 python tools/host_profile.py 2>/dev/null | tail -25 | tee -a gpurun_out/r6c11_host.txt
