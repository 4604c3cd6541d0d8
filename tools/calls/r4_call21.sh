# round 4, call 21: LDS-DMA of the filter k-tile issued at the start of the k-tile (scalar arithmetic in the MFMA shadow) vs at the end.
# A = HEAD, B = hoisted; kernel tests on B, then alternated bench runs on one box
cd /root/repo
cp abso/libvbg_B.so vibertgrid-pytorch_amd/libvbg.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -2
timeout 300 python tools/conv3_pw_bench.py 2>&1 | tail -12
for rep in 1 2 3; do for v in A B; do cp abso/libvbg_$v.so vibertgrid-pytorch_amd/libvbg.so; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('build $v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_us'])"; done; done
cp abso/libvbg_A.so vibertgrid-pytorch_amd/libvbg.so; timeout 300 python tools/conv3_pw_bench.py 2>&1 | tail -12
