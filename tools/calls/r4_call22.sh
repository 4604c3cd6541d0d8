# round 4, call 22: what bounds the fp16-pair NT ping-pong tiles -- A = HEAD, W1 = loop DMA made out-of-range (issued, moves nothing),
# W2 = no DMA in the loop, W3 = no fragment reads in the loop (results are garbage in W1-W3)
cd /root/repo
for v in A W1 W2 W3; do cp abso/libvbg_$v.so vibertgrid-pytorch_amd/libvbg.so; echo "== $v"; python tools/plane_pair_bench.py 30 2>&1 | grep -v amdgpu.ids; done
cp abso/libvbg_A.so vibertgrid-pytorch_amd/libvbg.so
