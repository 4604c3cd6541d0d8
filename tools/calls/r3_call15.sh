R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_ddp.py -m gpu -x -q 2>&1 | tail -25
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/call15_bench.json 2> gpurun_out/call15_bench.err; tail -c 4000 gpurun_out/call15_bench.json; tail -3 gpurun_out/call15_bench.err
for s in cfg3 cfg5; do timeout 600 python bench.py --shape $s --steps 5 --warmup 2 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>gpurun_out/call15_$s.err | cut -c1-700; tail -2 gpurun_out/call15_$s.err; done
