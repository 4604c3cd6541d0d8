# round 4, call 31: 64-deep k-tiles (full-line row pieces) for the fp16-pair NT 128 x 128 tile -- micro-benchmark against the 32-deep tiles
cd /root/repo
TILES=128129,128164,256128 python tools/plane_pair_bench.py 30 2>&1 | grep -v amdgpu.ids
TILES=128129,128164,256128 python tools/plane_pair_bench.py 30 2>&1 | grep -v amdgpu.ids
