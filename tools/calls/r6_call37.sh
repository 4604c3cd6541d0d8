cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r6c37_pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -1
AMP=1 MODEL=e2e KEEP=1 timeout 600 python tools/bwd_diff.py 2>/dev/null | cut -c1-200 > gpurun_out/r6c37_bwd_diff_amp.txt
AMP=0 MODEL=e2e KEEP=1 timeout 600 python tools/bwd_diff.py 2>/dev/null | cut -c1-200 > gpurun_out/r6c37_bwd_diff_fp32.txt
timeout 600 python tools/amp_repro.py 2>/dev/null | head -16 > gpurun_out/r6c37_amp_repro.txt
timeout 600 python tools/layer_entry_host.py 2>/dev/null > gpurun_out/r6c37_layer_entry_host.txt
B=8 timeout 600 python tools/layer_entry_host.py 2>/dev/null >> gpurun_out/r6c37_layer_entry_host.txt
