cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv3" 2>&1 | tail -3 | tee gpurun_out/r6c41_pytest.txt
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_full_scale.py tests/test_gpu_streams.py -x -q 2>&1 | tail -3 | tee -a gpurun_out/r6c41_pytest.txt
for i in 1 2; do timeout 300 python tools/infer_latency.py 2>/dev/null; done | tee gpurun_out/r6c41_infer.txt
