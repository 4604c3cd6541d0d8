# round 5, call 2: the RCCL one-rank test, the stock DDP test with its noise floor, the two new-fixture tests that exist so far, the phase
# profile of the stock loop (torch.optim / fused optimizers, pageable / resident inputs), the forced-reducer bench
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_ddp.py -q -m gpu --tb=short -rf -x -s -k "rccl or stock_ddp" 2>&1 | grep -v "^$" | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|pretrained will" | cut -c1-700 | tail -30
timeout 900 python -m pytest "tests/test_gpu_full_scale.py::test_full_scale_vs_reference_golden[cfg1]" "tests/test_gpu_full_scale.py::test_stock_loop_takes_the_all_pair_backward" -q -m gpu --tb=short -rf -s 2>&1 | grep -v "^$" | grep -v "amdgpu.ids\|pretrained will" | cut -c1-900 | tail -30
for o in torch fused; do python tools/stock_loop_profile.py --optim $o 2>/dev/null; python tools/stock_loop_profile.py --optim $o --resident 2>/dev/null; done
VBG_FORCE_REDUCER=1 python bench.py --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>gpurun_out/r5c2_forced.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced reducer:', d['value'], d['ms_per_step'], d['config'])"
tail -3 gpurun_out/r5c2_forced.err
