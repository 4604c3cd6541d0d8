cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/gemm_shapes_log.py 2>/dev/null > gpurun_out/r6c56_gemm_shapes.txt
wc -l gpurun_out/r6c56_gemm_shapes.txt
