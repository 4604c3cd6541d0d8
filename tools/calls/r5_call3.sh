# round 5, call 3: the RCCL one-rank test on its own, the forced-reducer bench with its exit code, homed vs plain stock loop printout,
# attention tests with the pooled masks, A/B of the mask pool, host profile of the step
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ddp.py -q -m gpu --tb=long -rf -x -s -k "rccl" 2>&1 | grep -v "^$" | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|pretrained will" | cut -c1-900 | tail -40
VBG_FORCE_REDUCER=1 timeout 600 python bench.py --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg > gpurun_out/r5c3_forced.json 2>gpurun_out/r5c3_forced.err; echo "forced bench rc=$?"
tail -5 gpurun_out/r5c3_forced.err | cut -c1-400; cut -c1-1500 gpurun_out/r5c3_forced.json
timeout 600 python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_attention.py -q -m gpu --tb=short -rf -s -k "stock_loop or attention" 2>&1 | grep "stock loop\|worst parameter\|passed\|failed\|Error" | cut -c1-600
bash tools/run_ab.sh VBG_MASK_POOL 2>&1 | grep "VBG_MASK_POOL=" 
python tools/host_profile.py --fp32 2>/dev/null | head -70
