# round 4, call 41: one counter hash per quad in the LayerNorm dropout (D) against one per element (A): tests, kernel micro-benchmark, step A/B
cd /root/repo
cp abso/libvbg_D.so vibertgrid-pytorch_amd/libvbg.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "ln or dropout or layernorm or bert or e2e" 2>&1 | tail -3
for v in A D A D; do cp abso/libvbg_$v.so vibertgrid-pytorch_amd/libvbg.so; echo "== $v"; python tools/ln_bwd_bench.py 2>/dev/null; done
for rep in 1 2 3; do for v in A D; do cp abso/libvbg_$v.so vibertgrid-pytorch_amd/libvbg.so; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('build $v', d['value'], d['ms_per_step'], d['config']['last_loss'])"; done; done
cp abso/libvbg_D.so vibertgrid-pytorch_amd/libvbg.so
