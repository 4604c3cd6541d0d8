# round 4, call 2: PW (pre-split filter) kernels: parity, shapes, A/B in the step; stock-DDP diagnostics
cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3" > gpurun_out/r4c2_conv3.txt 2>&1; echo "conv3 tests rc=$?"; tail -3 gpurun_out/r4c2_conv3.txt
python -m pytest tests/test_gpu_ddp.py -x -q -m gpu -s -k stock > gpurun_out/r4c2_ddp.txt 2>&1; echo "ddp rc=$?"; grep -n "losses stock\|running mean\|rel-L2 of the 3-step\|passed\|failed" gpurun_out/r4c2_ddp.txt | cut -c1-700
python tools/conv3_pw_bench.py > gpurun_out/r4c2_conv3_pw.txt 2>&1; echo "shapes rc=$?"; cat gpurun_out/r4c2_conv3_pw.txt | tail -60
for pw in 0 1 0 1; do VBG_CONV3_PW=$pw python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PW=$pw', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_us'])"; done
