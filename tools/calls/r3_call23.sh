R=$GRAFT_REPO_ROOT
cd $R
for cfgenv in "VBG_PAIR_BWD=0" "VBG_PAIR_BWD=1" "VBG_PAIR=0" "VBG_PAIR=0 VBG_CONV3_F16=0"; do
d=gpurun_out/dump_$(echo $cfgenv | tr ' =' '__')
mkdir -p $d
env $cfgenv VBG_DUMP_DIR=$d timeout 900 python -m pytest tests/test_gpu_full_scale.py -m gpu -q -k "cfg2e or cfg5e" 2>&1 | grep -E "parameter gradients|passed|failed" | cut -c1-200
done
