cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/conv3_small_sweep.py 2>&1 | grep -v Warning | tee gpurun_out/r6c43_conv3_small.txt
