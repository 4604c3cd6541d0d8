R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "plane_gemm or ln or split" 2>&1 | tail -6
mkdir -p gpurun_out/dump3
VBG_DUMP_DIR=gpurun_out/dump3 timeout 1200 python -m pytest tests/test_gpu_full_scale.py -m gpu -q 2>&1 | grep -E "AssertionError|Error|parameter gradients|class-prob|passed|failed" | cut -c1-300
bash tools/run_ab.sh VBG_PAIR 2>&1 | grep -v "^+" | tail -4
for t in 480 256; do VBG_CONV3_MIN_TILES_BWD=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MIN_TILES_BWD=$t', d['value'], d['ms_per_step'])"; done
for t in 480 256; do VBG_CONV3_MIN_TILES_BWD=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MIN_TILES_BWD=$t', d['value'], d['ms_per_step'])"; done
