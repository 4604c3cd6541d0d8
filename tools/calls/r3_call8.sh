R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 600 python tools/debug_segbias.py cfg5e 2>&1 | grep -v Warning | tail -12 > gpurun_out/call8_dbg5.txt
timeout 600 python tools/debug_segbias.py cfg4e 2>&1 | grep -v Warning | tail -12 > gpurun_out/call8_dbg4.txt
cat gpurun_out/call8_dbg5.txt gpurun_out/call8_dbg4.txt
VBG_TEST_PRECISION=fp32 VBG_DUMP_DIR=gpurun_out timeout 900 python -m pytest tests/test_gpu_full_scale.py -m gpu -q -k "cfg4e" 2>&1 | grep -E "AssertionError|parameter gradients|passed|failed" | cut -c1-400
