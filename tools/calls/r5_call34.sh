# round 5, call 34: the 64 -> 64 weight gradients of the first trunk stage on the row-reuse kernel's [64 x 9 x 64] blocks (fp16 form) vs the generic fp32-MFMA product
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r5c34_*.txt
for a in 0 1; do
  rm -rf /tmp/stepprof
  VBG_CONV3W_N64=$a timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stepprof -o e -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg > /dev/null 2>&1 < /dev/null
  echo "== VBG_CONV3W_N64=$a" >> gpurun_out/r5c34_w.txt
  python tools/kstat.py /tmp/stepprof "conv3x3_wgrad_kernel|gemm_kernel<64, 64, 32, 256, 1, 3|conv3_wgrad_reduce|amax_kernel" >> gpurun_out/r5c34_w.txt
done
cat gpurun_out/r5c34_w.txt
for i in 1 2; do for a in 1 0; do
  VBG_CONV3W_N64=$a timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n64=$a', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r5c34_ab.txt
VBG_CONV3W_N64=1 timeout 900 python -m pytest tests -m gpu -x -q -k "conv3 or full_scale_vs_reference_golden" > gpurun_out/r5c34_tests.txt 2>&1 < /dev/null
tail -2 gpurun_out/r5c34_tests.txt
