cd /root/repo
mkdir -p gpurun_out
KEEP=1 AMP=1 MODEL=e2e timeout 600 python tools/bwd_diff.py 2>&1 | grep -v "Warning\|warn" | cut -c1-260 | tail -150 | tee gpurun_out/r6c36_bwd_diff_e2e_amp.txt
