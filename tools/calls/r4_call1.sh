# round 4, call 1: the new parity tests (cfg3 / cfg3e / cfg2e8 full scale, stock DDP wrapper) and a first bench line
cd /root/repo
python -m pytest tests/test_gpu_full_scale.py -x -q -m gpu -k "cfg3 or cfg2e8" -s > gpurun_out/r4c1_full.txt 2>&1; echo "full rc=$?" 
python -m pytest tests/test_gpu_ddp.py -x -q -m gpu -s > gpurun_out/r4c1_ddp.txt 2>&1; echo "ddp rc=$?"
python bench.py --steps 10 --warmup 3 > gpurun_out/r4c1_bench.json 2> gpurun_out/r4c1_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r4c1_full.txt; tail -3 gpurun_out/r4c1_ddp.txt; cut -c1-600 gpurun_out/r4c1_bench.json
