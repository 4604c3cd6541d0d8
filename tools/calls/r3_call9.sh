R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv or bn_" 2>&1 | tail -4
timeout 600 python tools/conv3_forms_bench.py 2>&1 | grep -E "wgrad|fwd   f16|dgrad f16x2 \(amax" > gpurun_out/call9_conv3_forms.txt
cat gpurun_out/call9_conv3_forms.txt
mkdir -p gpurun_out/dump2
VBG_DUMP_DIR=gpurun_out/dump2 timeout 1200 python -m pytest tests/test_gpu_full_scale.py tests/test_gpu_model.py -m gpu -q 2>&1 | grep -E "AssertionError|parameter gradients|passed|failed" | cut -c1-400
bash tools/run_ab.sh VBG_CONV3_F16_BWD 2>&1 | grep -v "^+" | tail -4
