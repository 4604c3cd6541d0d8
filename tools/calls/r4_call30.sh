# round 4, call 30: amp -- attention-output projection on the hi plane of O; amp tests; amp bench
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_full_scale.py tests/test_gpu_model.py -x -q -m gpu -k "amp" -s 2>&1 | grep "amp:\|passed\|failed"
for v in 1 2; do python bench.py --amp --steps 20 --warmup 5 --no-cpu-baseline --no-h2d-leg 2>gpurun_out/call30.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('amp', d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/amp_prof -o amp -- python /root/repo/bench.py --amp --steps 5 --warmup 2 --no-cpu-baseline --no-h2d-leg > /dev/null 2>&1
cd /root/repo; ls gpurun_out/amp_prof | head
