# round 5, call 9: the forced one-rank RCCL reducer: what do SyncBatchNorm statistics collectives and the bucket machinery cost per step?
cd /root/repo
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg"
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain:', d['value'], d['ms_per_step'])"
for extra in "" "--no-syncbn" "--no-ddp-overlap" "--syncbn-comm own"; do VBG_FORCE_REDUCER=1 $B $extra 2>gpurun_out/r5c9_forced.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced reducer $extra:', d['value'], d['ms_per_step'], d['config'].get('syncbn_collectives'), d['config'].get('parallelism'), d['config'].get('ddp_overlap'))" || tail -5 gpurun_out/r5c9_forced.err; done
