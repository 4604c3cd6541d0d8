# round 3, GPU call 3: ping-pong probe with the wide-request what-if and the 256x128 two-stage ping-pong; colsum_f64 + cfg4e/cfg5e tests
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 600 $R/tools/probes/pingpong_gemm_probe > $R/gpurun_out/pingpong_probe2.txt 2>&1
cd $R
timeout 900 python -m pytest tests/test_gpu_full_scale.py tests/test_gpu_kernels.py -m gpu -q -k "cfg4e or cfg5e or colsum" 2>&1 | tail -30 > gpurun_out/call3_pytest.txt
cat gpurun_out/pingpong_probe2.txt gpurun_out/call3_pytest.txt
