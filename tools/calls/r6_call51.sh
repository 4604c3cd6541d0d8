cd /root/repo
mkdir -p gpurun_out
for i in 1 2; do
for v in 1 0; do
VBG_LAYER_ENTRY=$v python bench.py --amp --steps 20 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null | grep "^{" > /tmp/a.json
python -c "import json; d=json.load(open('/tmp/a.json')); print('entry$v amp standalone', d['value'], d['ms_per_step'], d.get('host'))" | tee -a gpurun_out/r6c51_amp.txt
done
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null | grep "^{" > /tmp/b.json
python -c "import json; d=json.load(open('/tmp/b.json')); print('default', d['value'], d['ms_per_step'], 'amp leg', d['amp']['value'], d['amp'].get('ms_per_step'))" | tee -a gpurun_out/r6c51_amp.txt
