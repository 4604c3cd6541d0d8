cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_abi.py -x -q -k "fp16_pair_form_of_the_forward or gemm or slab or abi or offsets" 2>&1 | grep -E "^E |passed|failed|Error" | head -20 | tee gpurun_out/r6c63_pytest.txt
: > gpurun_out/r6c63_ab.txt
for i in 1 2 3; do
  for v in 0 1; do
    VBG_GEMM_F16=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null | grep "^{" > /tmp/x.json
    python -c "import json; d=json.load(open('/tmp/x.json')); print('gemm_f16 $v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r6c63_ab.txt
  done
done
