cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "slab or gemm" 2>&1 | tail -3 | tee gpurun_out/r6c46_pytest.txt
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_layer_entry.py tests/test_gpu_full_scale.py tests/test_gpu_heads.py -x -q 2>&1 | tail -3 | tee -a gpurun_out/r6c46_pytest.txt
for v in 0 1 0 1; do echo "VBG_SLAB_SPLIT=$v"; VBG_SLAB_SPLIT=$v VBG_INFER_BATCHES=1 timeout 300 python tools/infer_latency.py 2>/dev/null; done | tee gpurun_out/r6c46_infer.txt
timeout 300 python tools/infer_latency.py 2>/dev/null | tee -a gpurun_out/r6c46_infer.txt
