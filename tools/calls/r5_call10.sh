# round 5, call 10: direct RCCL communicator for the SyncBatchNorm statistics: the one-rank test, then the forced-reducer bench in all modes
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ddp.py -q -m gpu --tb=short -rf -x -s -k "rccl" 2>&1 | grep "RCCL one-rank\|passed\|failed\|Error\|error\|assert" | cut -c1-1200
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg"
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain:', d['value'], d['ms_per_step'])"
for extra in "" "--syncbn-comm shared" "--syncbn-comm shared --no-ddp-overlap" "--no-syncbn" "--no-ddp-overlap"; do VBG_FORCE_REDUCER=1 $B $extra 2>gpurun_out/r5c10_forced.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced reducer $extra:', d['value'], d['ms_per_step'], d['config'].get('syncbn_collectives'), d['config'].get('syncbn_comm'), d['config'].get('ddp_overlap'))" || tail -5 gpurun_out/r5c10_forced.err; done
