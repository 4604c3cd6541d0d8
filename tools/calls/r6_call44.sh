cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/small_gemm_cold.py 2>&1 | grep -v Warning | tee gpurun_out/r6c44_small_gemm_cold.txt
