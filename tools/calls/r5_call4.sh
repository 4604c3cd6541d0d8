# round 5, call 4: RCCL test again, the stock DDP test with homing on / off (is the 3-step distance the homing or the fixture's chaos?),
# the two batch fixtures, ResCell A/B, forced reducer with / without SyncBatchNorm
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ddp.py -q -m gpu --tb=short -rf -x -s -k "rccl" 2>&1 | grep "RCCL one-rank\|passed\|failed\|Error\|assert" | cut -c1-900
for h in 0 1 0 1; do echo "VBG_HOME=$h"; VBG_HOME=$h timeout 600 python -m pytest tests/test_gpu_ddp.py -q -m gpu --tb=line -s -k "stock_ddp" 2>&1 | grep "losses stock\|three steps, worst\|run-to-run\|passed\|failed" | cut -c1-330; done
timeout 1500 python -m pytest "tests/test_gpu_full_scale.py::test_full_scale_every_gradient_vs_reference[cfg4e8]" "tests/test_gpu_full_scale.py::test_full_scale_every_gradient_vs_reference[cfg5e16]" -q -m gpu --tb=short -rf -s 2>&1 | grep -v "^$" | grep -v "amdgpu.ids\|pretrained will\|Warning\|warn" | cut -c1-1000 | tail -30
bash tools/run_ab.sh VBG_RES_CELL 2>&1 | grep "VBG_RES_CELL="
for extra in "" "--no-syncbn"; do VBG_FORCE_REDUCER=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced reducer $extra:', d['value'], d['ms_per_step'], d['config'].get('syncbn_collectives'), d['config'].get('parallelism'))"; done
