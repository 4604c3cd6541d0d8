cd /root/repo; mkdir -p gpurun_out
timeout 300 python tools/infer_host_profile.py > gpurun_out/r6c12_infer_host.txt 2>&1
timeout 300 python tools/infer_latency.py >> gpurun_out/r6c12_infer_host.txt 2>&1
head -70 gpurun_out/r6c12_infer_host.txt | cut -c1-180
