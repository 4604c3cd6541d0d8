set -x
mkdir -p gpurun_out/ab
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg"
for i in 1 2 3; do for x in 1 2; do env VBG_WGRAD_STREAM=$x $B 2>gpurun_out/call29.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VBG_WGRAD_STREAM=$x', d['value'], d['ms_per_step'], d['config']['last_loss'])" | tee -a gpurun_out/ab/wgrad_stream.txt; done; done
python tools/step_gemm_profile.py > gpurun_out/step_gemm.txt 2>gpurun_out/step_gemm.err
tail -5 gpurun_out/step_gemm.err
