# round 5, call 37: the encoder on its own stream beside the CNN's first stage (VBG_OVERLAP=1) as the default? full GPU suite with it on, every bench leg A/B
cd /root/repo
mkdir -p gpurun_out
VBG_OVERLAP=1 timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5c37_tests.txt 2>&1 < /dev/null
tail -2 gpurun_out/r5c37_tests.txt
for a in 1 0; do
  VBG_OVERLAP=$a timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null < /dev/null > gpurun_out/r5c37_bench_$a.json
  python -c "
import json; d=json.load(open('gpurun_out/r5c37_bench_$a.json'))
print('overlap=$a', d['value'], d['ms_per_step'], 'h2d', d['h2d_inclusive']['value'], 'stock', d['stock_loop']['value'], d['stock_loop']['resident_inputs']['value'], 'amp', d['amp']['value'], 'strict', d['bf16x3_strict']['value'], 'roofline', d['roofline']['frac'], d['roofline_conv3']['frac'])"
done | tee gpurun_out/r5c37_ab.txt
VBG_OVERLAP=1 timeout 300 python tools/infer_latency.py 2>/dev/null | tee gpurun_out/r5c37_infer_1.txt
VBG_OVERLAP=0 timeout 300 python tools/infer_latency.py 2>/dev/null | tee gpurun_out/r5c37_infer_0.txt
VBG_OVERLAP=1 VBG_FORCE_REDUCER=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced reducer, overlap=1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5c37_ab.txt
VBG_OVERLAP=0 VBG_FORCE_REDUCER=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced reducer, overlap=0', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5c37_ab.txt
