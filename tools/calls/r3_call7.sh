R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 600 python tools/conv3_forms_bench.py > gpurun_out/call7_conv3_forms.txt 2>&1
cat gpurun_out/call7_conv3_forms.txt | grep -v "32x32\|64x64 128"
for x in 0 1; do
mkdir -p gpurun_out/dump$x
VBG_DUMP_DIR=gpurun_out/dump$x VBG_CONV3_F16_BWD=$x timeout 900 python -m pytest tests/test_gpu_full_scale.py -m gpu -q -k "cfg4e or cfg2e or cfg5e" 2>&1 | grep -E "AssertionError|parameter gradients|passed|failed" | cut -c1-600
done
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv3x3 or bn_" 2>&1 | tail -2
