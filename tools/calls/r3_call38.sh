python -m pytest tests/test_gpu_kernels.py -q -x -k "plane_gemm or pair" 2>&1 | tail -3
for v in 0 1; do echo "VBG_PAIR_DEEP=$v"; VBG_PAIR_DEEP=$v python tools/plane_gemm_bench.py 2>&1 | grep -i "pair\|f16" | head -24; done
bash tools/run_ab.sh VBG_PAIR_DEEP 2>&1 | grep -v "^+"
