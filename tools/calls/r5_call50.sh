# round 5, call 50: opt-in -- the encoder's optimizer group living on the side stream end to end: three steps against the default mode, and the bench leg
cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/pipeline_check.py > gpurun_out/r5c50_pipe.txt 2> gpurun_out/r5c50_pipe.err < /dev/null
cat gpurun_out/r5c50_pipe.txt | cut -c1-330; tail -3 gpurun_out/r5c50_pipe.err | cut -c1-300
for i in 1 2; do
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --pipelined-leg 2>gpurun_out/r5c50_bench.err < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], 'pipelined', d.get('pipelined_encoder_group'))"
done | tee gpurun_out/r5c50_ab.txt
tail -3 gpurun_out/r5c50_bench.err | cut -c1-300
