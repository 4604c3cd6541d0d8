# round 4, call 6: wide-grid column sums of the pair split: kernel tests, A/B in the step, kernel stats
cd /root/repo
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_attention.py -x -q -m gpu > gpurun_out/r4c6_kernels.txt 2>&1; echo "kernel tests rc=$?"; tail -3 gpurun_out/r4c6_kernels.txt
for ws in 0 1 0 1; do VBG_COLSUM_WS=$ws python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('COLSUM_WS=$ws', d['value'], d['ms_per_step'])"; done
TOPN=12 bash tools/prof_step.sh 2>&1 | grep -E "split_planes|kernel ms|launches/step"
