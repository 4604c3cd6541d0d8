# round 5, call 48: side-stream forks as one C call + set_stream, operands held until the backward ends (VBG_LIGHT_FORK): full suite, race check, host enqueue time and legs A/B
cd /root/repo
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r5c48_tests.txt 2>&1 < /dev/null
tail -2 gpurun_out/r5c48_tests.txt
timeout 600 python tools/stream_race_check.py --reps 3 2>/dev/null > gpurun_out/r5c48_race.txt < /dev/null; tail -4 gpurun_out/r5c48_race.txt | cut -c1-200
for a in 1 0 1 0; do
  VBG_LIGHT_FORK=$a timeout 300 python tools/stock_loop_profile.py --optim torch 2>/dev/null < /dev/null | grep -E "per step|backward: host|forward: host" | tr '\n' ' ' | sed "s/^/light_fork=$a /"; echo
done | tee gpurun_out/r5c48_host.txt
for a in 1 0 1 0; do
  VBG_LIGHT_FORK=$a timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('light_fork=$a', d['value'], d['ms_per_step'], 'stock', d['stock_loop']['value'], 'amp', d['amp']['value'])"
done | tee gpurun_out/r5c48_ab.txt
