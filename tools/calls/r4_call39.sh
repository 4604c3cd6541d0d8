# round 4, call 39: rows per wave of the LayerNorm backward (A: 2, L4: 4, L8: 8)
cd /root/repo
for v in A L4 L8 A L4 L8; do cp abso/libvbg_$v.so vibertgrid-pytorch_amd/libvbg.so; echo "== $v"; python tools/ln_bwd_bench.py 2>/dev/null; done
cp abso/libvbg_A.so vibertgrid-pytorch_amd/libvbg.so
