# round 3, GPU call 4: ping-pong k-loop in the shipped NT plane kernel: kernel tests, per-shape times both ways, step A/B
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_attention.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/call4_pytest.txt
cat gpurun_out/call4_pytest.txt
for x in 0 1; do VBG_PINGPONG=$x timeout 300 python tools/plane_gemm_bench.py --tiles 128129,128130,256128 > gpurun_out/call4_pg_$x.txt 2>&1; done
paste -d'\n' gpurun_out/call4_pg_0.txt gpurun_out/call4_pg_1.txt | head -80
bash tools/run_ab.sh VBG_PINGPONG 2>&1 | grep -v "^+" | tail -8
