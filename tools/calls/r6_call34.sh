cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/amp_repro.py 2>/dev/null | tee gpurun_out/r6c34_amp_repro.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o inf -- python /root/repo/tools/infer_latency.py > /tmp/inf.log 2>&1
tail -3 /tmp/inf.log
f=$(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1); cp "$f" /root/repo/gpurun_out/r6c34_infer_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r6c34_infer_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('total kernel ms', tot/1e6, 'launches', calls)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:30]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms  {r['Name'][:100]}")
PY
