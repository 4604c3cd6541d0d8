# round 5, call 11: the whole data-parallel test file after the SyncBatchNorm route change (fold + count in one launch, finalize in the apply
# kernel), the forced-reducer bench again
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_ddp.py -q -m gpu --tb=short -rf 2>&1 | tail -4
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg"
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain:', d['value'], d['ms_per_step'])"
for extra in "" "--syncbn-comm shared" "--no-syncbn"; do VBG_FORCE_REDUCER=1 $B $extra 2>gpurun_out/r5c11_forced.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced reducer $extra:', d['value'], d['ms_per_step'], d['config'].get('syncbn_collectives'), d['config'].get('syncbn_comm'), d['config'].get('ddp_overlap'))" || tail -5 gpurun_out/r5c11_forced.err; done
