cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "small_tile or conv3" 2>&1 | tail -3 | tee gpurun_out/r6c45_pytest.txt
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_layer_entry.py tests/test_gpu_full_scale.py -x -q 2>&1 | tail -3 | tee -a gpurun_out/r6c45_pytest.txt
for i in 1 2; do timeout 300 python tools/infer_latency.py 2>/dev/null; done | tee gpurun_out/r6c45_infer.txt
bash tools/calls/r6_call42.sh > /dev/null 2>&1
cp gpurun_out/r6c42_infer_b1_trace.txt gpurun_out/r6c45_infer_b1_trace.txt
tail -1 gpurun_out/r6c45_infer_b1_trace.txt
