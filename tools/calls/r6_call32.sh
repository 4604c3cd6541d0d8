cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_layer_entry.py -x -q -k "amp" 2>&1 | grep -v Warning | tail -40 | tee gpurun_out/r6c32_pytest.txt
timeout 600 python tools/layer_entry_host.py 2>/dev/null | tee gpurun_out/r6c32_host.txt
