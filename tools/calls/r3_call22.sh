R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/dump5
VBG_DUMP_DIR=gpurun_out/dump5 timeout 1500 python -m pytest tests/test_gpu_full_scale.py tests/test_gpu_model.py -m gpu -q 2>&1 | grep -E "Error|error|parameter gradients|class-prob|passed|failed|assert" | cut -c1-500 | tail -30
