# round 5, call 33: the stem's patch matrix and filter padded to whole k-tiles (aligned operands, split form): tests, kernel times, A/B of the step
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "backbone or e2e or full_scale or train_loop or conv or amp or inference" > gpurun_out/r5c33_tests.txt 2>&1 < /dev/null
tail -2 gpurun_out/r5c33_tests.txt
for a in 1 0; do
  rm -rf /tmp/stepprof
  VBG_STEM_ALIGNED=$a timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stepprof -o e -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg > /dev/null 2>&1 < /dev/null
  echo "== VBG_STEM_ALIGNED=$a" >> gpurun_out/r5c33_stem.txt
  python tools/kstat.py /tmp/stepprof "im2col|gemm_kernel<64, 64, (16|32), 256, (0, 0|1, 1), (true|false), (0|3)>" >> gpurun_out/r5c33_stem.txt
done
cat gpurun_out/r5c33_stem.txt
for i in 1 2; do for a in 1 0; do
  VBG_STEM_ALIGNED=$a timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('aligned=$a', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r5c33_ab.txt
