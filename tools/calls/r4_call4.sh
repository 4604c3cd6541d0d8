# round 4, call 4: software-pipelined PW convolution kernels: parity, shapes, A/B in the step
cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3" > gpurun_out/r4c4_conv3.txt 2>&1; echo "conv3 tests rc=$?"; tail -3 gpurun_out/r4c4_conv3.txt
VBG_CONV3_PIPE=0 python tools/conv3_pw_bench.py > gpurun_out/r4c4_pw_lockstep.txt 2>&1
python tools/conv3_pw_bench.py > gpurun_out/r4c4_pw_pipe.txt 2>&1; echo "shapes rc=$?"
paste -d'|' <(grep -E "PW " gpurun_out/r4c4_pw_lockstep.txt | cut -c1-75) <(grep -E "PW " gpurun_out/r4c4_pw_pipe.txt | cut -c52-80)
tail -1 gpurun_out/r4c4_pw_pipe.txt
for pp in 0 1 0 1; do VBG_CONV3_PIPE=$pp python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PIPE=$pp', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_us'])"; done
