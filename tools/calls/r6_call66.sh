cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_attention.py -x -q 2>&1 | grep -E "^E |passed|failed" | head -8 | tee gpurun_out/r6c66_pytest.txt
timeout 300 python tools/attn_bench.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tee gpurun_out/r6c66_attn.txt
