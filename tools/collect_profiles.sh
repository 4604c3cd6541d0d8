# every rocprofv3 collection the committed profiles/ are made from, on one box:  gpurun -- bash tools/collect_profiles.sh ; python tools/make_profiles.py r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass"
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/bench_line.err | grep "^{" > gpurun_out/bench_line.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e -o e -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass > gpurun_out/prof_e.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_amp -o amp -- python bench.py --amp --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-stock-leg --no-single-stream-pass > gpurun_out/prof_amp.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- $B > gpurun_out/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- $B > gpurun_out/pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -o m -- $B > gpurun_out/pmc_m.log 2>&1
timeout 600 python tools/step_conv3_profile.py 2>/dev/null > gpurun_out/step_conv3.txt
timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/attn_shapes.txt
timeout 300 python tools/step_plane_profile.py 2>/dev/null > gpurun_out/step_plane.txt
timeout 600 python bench.py --amp --steps 20 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-stock-leg 2>/dev/null | grep "^{" > gpurun_out/bench_amp_line.json
rm -f gpurun_out/prof_amp/amp_kernel_trace.csv
ls -la gpurun_out/prof_e gpurun_out/pmc_f gpurun_out/pmc_m | head -30; cut -c1-400 gpurun_out/bench_line.json
# exploratory shapes and legs of the same build
for s in cfg3 cfg4 cfg5; do timeout 600 python bench.py --shape $s --steps 6 --warmup 2 --no-cpu-baseline --no-h2d-leg --no-stock-leg 2>/dev/null | grep "^{" > gpurun_out/bench_$s.json; done
timeout 300 python tools/infer_latency.py > gpurun_out/infer_latency.txt 2>&1
# round 5: the reference's loop phase by phase (torch.optim / fused optimizers, pageable / resident inputs), the one-rank RCCL reducer
( for o in torch fused; do python tools/stock_loop_profile.py --optim $o 2>/dev/null; python tools/stock_loop_profile.py --optim $o --resident 2>/dev/null; done ) > gpurun_out/stock_loop_phases.txt
( BB="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg"
  $BB 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one process, no process group:', d['value'], 'docs/s', d['ms_per_step'], 'ms')"
  for extra in "" "--syncbn-comm shared" "--no-syncbn" "--no-ddp-overlap"; do VBG_FORCE_REDUCER=1 $BB $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('one-rank RCCL group, FlatReducer + SyncBatchNorm $extra:', d['value'], 'docs/s', d['ms_per_step'], 'ms;', c.get('syncbn_collectives'), 'statistics collectives;', c.get('syncbn_comm'), '; overlap', c.get('ddp_overlap'), '; buckets', c.get('buckets'))"; done ) > gpurun_out/forced_reducer.txt
timeout 900 python tools/stream_race_check.py --reps 24 --only-default 2>/dev/null | grep -v "noise floor #" > gpurun_out/stream_race_final.txt
