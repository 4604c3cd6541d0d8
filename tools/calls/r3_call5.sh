# round 3, GPU call 5: fp16-form input gradients of the wide 3x3 convolutions (dy scaled by its largest magnitude): tests + step A/B
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv3x3 or bn_ or colsum or conv" 2>&1 | tail -8 > gpurun_out/call5_pytest.txt
cat gpurun_out/call5_pytest.txt
timeout 1200 python -m pytest tests/test_gpu_full_scale.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/call5_pytest2.txt
cat gpurun_out/call5_pytest2.txt
bash tools/run_ab.sh VBG_CONV3_F16_BWD 2>&1 | grep -v "^+" | tail -8
