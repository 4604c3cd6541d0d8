R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/dump4
VBG_DUMP_DIR=gpurun_out/dump4 timeout 1500 python -m pytest tests/test_gpu_full_scale.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | grep -E "Error|error|parameter gradients|class-prob|passed|failed|assert" | cut -c1-400 | tail -30
bash tools/run_ab.sh VBG_PAIR_BWD 2>&1 | grep -v "^+" | tail -4
