# round 5, call 12: host-device synchronisation points of the step (torch's sync debug mode), plain and with the forced one-rank reducer
cd /root/repo
python tools/sync_points.py 2>&1 | grep -v "amdgpu.ids\|pretrained will\|socket.cpp" | cut -c1-400 | tail -30
python tools/sync_points.py --forced 2>&1 | grep -v "amdgpu.ids\|pretrained will\|socket.cpp\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | cut -c1-400 | tail -30
