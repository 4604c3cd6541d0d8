# round 3, GPU call 2: LDS-DMA ingest microbenchmark, ping-pong schedule probe, the failing cfg4e gradient test
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 300 $R/tools/probes/dma_rate_probe > $R/gpurun_out/dma_rate.txt 2>&1
timeout 600 $R/tools/probes/pingpong_gemm_probe > $R/gpurun_out/pingpong_probe.txt 2>&1
cd $R
timeout 900 python -m pytest tests/test_gpu_full_scale.py -m gpu -q -k "cfg4e or cfg5e" 2>&1 | tail -60 > gpurun_out/call2_pytest.txt
cat gpurun_out/dma_rate.txt gpurun_out/pingpong_probe.txt gpurun_out/call2_pytest.txt
