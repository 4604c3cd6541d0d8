# round 4, call 20: three builds of conv3.hip on one box -- A: reduction order (kh, chunk, kw) static (HEAD), B: (chunk pair, kh, chunk, kw)
# static, C: run-time switch -- alternated
cd /root/repo
cp vibertgrid-pytorch_amd/libvbg.so /tmp/libvbg_keep.so
for rep in 1 2; do for v in A B C; do cp abso/libvbg_$v.so vibertgrid-pytorch_amd/libvbg.so; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('build $v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_us'])"; done; done
cp /tmp/libvbg_keep.so vibertgrid-pytorch_amd/libvbg.so
