cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_layer_entry.py -q 2>&1 | grep -v Warning | grep -E "^E  |passed|failed|FAILED" | head -30 | tee gpurun_out/r6c33_pytest.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o inf -- python /root/repo/tools/infer_latency.py > /tmp/inf.log 2>&1
tail -3 /tmp/inf.log
f=$(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1); cp "$f" /root/repo/gpurun_out/r6c33_infer_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r6c33_infer_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('total kernel ms', tot/1e6, 'launches', calls)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:25]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms  {r['Name'][:100]}")
PY
