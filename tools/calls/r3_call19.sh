R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "plane_gemm or split" 2>&1 | tail -25
