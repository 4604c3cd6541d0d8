R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 600 python tools/conv3_forms_bench.py > gpurun_out/call6_conv3_forms.txt 2>&1
cat gpurun_out/call6_conv3_forms.txt
timeout 900 python -m pytest tests/test_gpu_full_scale.py -m gpu -x -q -k cfg4e 2>&1 | grep -E "AssertionError|cfg4e:|passed|failed" | cut -c1-1500 > gpurun_out/call6_pytest.txt
cat gpurun_out/call6_pytest.txt
