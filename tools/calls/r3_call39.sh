python -m pytest tests/test_gpu_kernels.py -q -x -k "plane_gemm or pair" 2>&1 | tail -3
bash tools/run_ab.sh VBG_PAIR_WIDE 2>&1 | grep -v "^+"
