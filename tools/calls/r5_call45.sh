# round 5, call 45: the FINAL tree (two-stream defaults, 64-filter weight gradients on the row-reuse kernel): every collection profiles/r05_* are made from, on one box
cd /root/repo
mkdir -p gpurun_out
( time bash tools/collect_profiles.sh ) 2>&1 < /dev/null | tail -30 | cut -c1-600
cat gpurun_out/stock_loop_phases.txt gpurun_out/forced_reducer.txt gpurun_out/infer_latency.txt gpurun_out/stream_race.txt 2>/dev/null | cut -c1-300
du -sh gpurun_out
