# round 6, call 30: host profile of the batch-1 inference call
cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/infer_profile.py > gpurun_out/r6c30_infer_profile.txt 2>/dev/null
head -80 gpurun_out/r6c30_infer_profile.txt
