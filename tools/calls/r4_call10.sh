# round 4, call 10: the rest of the -m gpu suite (everything behind the stock-DDP test), the LDS-staged RoIAlign A/B
cd /root/repo
python -m pytest tests -q -m gpu > gpurun_out/r4c10_pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r4c10_pytest.txt
for v in 0 1 0 1; do VBG_ROI_LDS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ROI_LDS=$v', d['value'], d['ms_per_step'])"; done
TOPN=60 bash tools/prof_step.sh 2>&1 | grep -E "roi_align|kernel ms"
