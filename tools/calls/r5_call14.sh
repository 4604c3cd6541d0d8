# round 5, call 14: every collection the committed profiles/r05_* are made from (tools/collect_profiles.sh), on one box
cd /root/repo
( time bash tools/collect_profiles.sh ) 2>&1 | tail -30 | cut -c1-600
cat gpurun_out/stock_loop_phases.txt gpurun_out/forced_reducer.txt gpurun_out/infer_latency.txt 2>/dev/null | cut -c1-300
ls -la gpurun_out/prof_e gpurun_out/pmc_m | head; du -sh gpurun_out
