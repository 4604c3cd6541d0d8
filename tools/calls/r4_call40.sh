# round 4, call 40: branch-free erf in the GELU epilogues (E) against ocml's erff (A): tests, epilogue micro-benchmark, step A/B
cd /root/repo
cp abso/libvbg_E.so vibertgrid-pytorch_amd/libvbg.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gelu or pair_form or bound_scaled or plane_gemm" 2>&1 | tail -3
for v in A E A E; do cp abso/libvbg_$v.so vibertgrid-pytorch_amd/libvbg.so; echo "== $v"; python tools/plane_epi_bench.py 2>/dev/null | grep "256128" | grep "plain fp32 store$\|FFN1\|bound"; done
for rep in 1 2 3; do for v in A E; do cp abso/libvbg_$v.so vibertgrid-pytorch_amd/libvbg.so; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('build $v', d['value'], d['ms_per_step'], d['config']['last_loss'])"; done; done
cp abso/libvbg_E.so vibertgrid-pytorch_amd/libvbg.so
