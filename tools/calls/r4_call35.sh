# round 4, call 35: AdamW kernel with two quads per thread in flight (B) against one (A)
cd /root/repo
for v in A B A B; do cp abso/libvbg_$v.so vibertgrid-pytorch_amd/libvbg.so; echo -n "$v "; python tools/adamw_bench.py 2>/dev/null; done
cp abso/libvbg_B.so vibertgrid-pytorch_amd/libvbg.so
timeout 600 python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_kernels.py -x -q -m gpu -k "adam or optim or sgd" 2>&1 | tail -2
