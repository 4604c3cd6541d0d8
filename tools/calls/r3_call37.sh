python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -q -x -k "loss or e2e or model or ohem or sample" 2>&1 | tail -3
bash tools/run_ab.sh VBG_ASYNC_H2D 2>&1 | grep -v "^+"
