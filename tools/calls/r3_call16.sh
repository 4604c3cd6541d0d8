R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_ddp.py -m gpu -x -q -k "r34 or unequal" 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo\|in evaluation" | tail -30
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/call16_bench.json 2> gpurun_out/call16_bench.err; tail -2 gpurun_out/call16_bench.err
