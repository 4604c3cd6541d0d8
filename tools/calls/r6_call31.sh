# round 6, call 31: the one-call encoder layer: bit-identity tests, encoder / model / full-scale parity, inference latency and the bench A/B
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_layer_entry.py -x -q 2>&1 | tail -15 | tee gpurun_out/r6c31_pytest.txt
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_full_scale.py -x -q 2>&1 | tail -5 | tee -a gpurun_out/r6c31_pytest.txt
for v in 0 1; do echo "VBG_LAYER_ENTRY=$v"; VBG_LAYER_ENTRY=$v timeout 300 python tools/infer_latency.py 2>/dev/null; done | tee gpurun_out/r6c31_infer.txt
: > gpurun_out/r6c31_ab.txt
for i in 1 2; do
  for v in 0 1; do
    VBG_LAYER_ENTRY=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-single-stream-pass > /tmp/b.json 2>/dev/null
    python -c "import json; d=json.load(open('/tmp/b.json')); print('entry$v', d['value'], d['ms_per_step'], d['stock_loop']['value'], d.get('host'))" | tee -a gpurun_out/r6c31_ab.txt
  done
done
