# round 4, call 7: 4-wave two-per-CU tile of the fp16-pair NT products against the 8-wave ping-pong tiles; A/B in the step
cd /root/repo
python tools/pair_tile_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c7_pair_tiles.txt
for t4 in 0 1 2 0 1 2; do VBG_PAIR_TILE4=$t4 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TILE4=$t4', d['value'], d['ms_per_step'], d['roofline_nt']['frac'], d['roofline_nt']['avg_us'])"; done
