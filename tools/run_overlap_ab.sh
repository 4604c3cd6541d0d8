set -x
mkdir -p gpurun_out/ov
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg"
VBG_OVERLAP=0 $B > gpurun_out/ov/b_off.json 2> gpurun_out/ov/b_off.err
VBG_OVERLAP=1 VBG_WGRAD_STREAM=0 $B > gpurun_out/ov/b_side.json 2> gpurun_out/ov/b_side.err
VBG_OVERLAP=1 VBG_WGRAD_STREAM=1 $B > gpurun_out/ov/b_both.json 2> gpurun_out/ov/b_both.err
VBG_OVERLAP=0 $B > gpurun_out/ov/b_off2.json 2> gpurun_out/ov/b_off2.err
tail -n 3 gpurun_out/ov/*.json gpurun_out/ov/*.err
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/ov/pytest.log
cat gpurun_out/ov/pytest.log
