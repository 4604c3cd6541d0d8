R=$GRAFT_REPO_ROOT
cd $R
python tools/bn_amax_bench.py 2>&1 | grep -v amdgpu
TOPN=12 bash tools/prof_step.sh 2>&1 | grep -E "amax|bn_|kernel ms|launches"
