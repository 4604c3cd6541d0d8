# round 5, call 31: the final tree -- full GPU suite, smoke, then every collection the committed profiles/r05_* are made from
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5c31_tests.txt 2>&1 < /dev/null
tail -2 gpurun_out/r5c31_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5c31_smoke.txt 2>&1 < /dev/null
tail -1 gpurun_out/r5c31_smoke.txt
( time bash tools/collect_profiles.sh ) 2>&1 < /dev/null | tail -30 | cut -c1-600
cat gpurun_out/stock_loop_phases.txt gpurun_out/forced_reducer.txt gpurun_out/infer_latency.txt 2>/dev/null | cut -c1-300
du -sh gpurun_out
