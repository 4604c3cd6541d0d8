R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/dump6
VBG_DUMP_DIR=gpurun_out/dump6 timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "Error|parameter gradients|passed|failed|^FAILED" | cut -c1-300 | tail -20
bash tools/run_ab.sh VBG_PAIR_BWD 2>&1 | grep -v "^+" | tail -4
