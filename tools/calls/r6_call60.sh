cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/gemm_f16_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6c60_gemm_f16.txt
