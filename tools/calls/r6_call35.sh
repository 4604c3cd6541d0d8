cd /root/repo
mkdir -p gpurun_out
ONE_STREAM=1 timeout 600 python tools/amp_repro.py 2>/dev/null | tee gpurun_out/r6c35_amp_repro_one_stream.txt
timeout 1500 python tools/poison_check.py --amp 2>/dev/null | tee gpurun_out/r6c35_poison_amp.txt
cd /tmp && export TMPDIR=/tmp
VBG_INFER_BATCHES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o inf -- python /root/repo/tools/infer_latency.py > /tmp/inf.log 2>&1
f=$(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1); cp "$f" /root/repo/gpurun_out/r6c35_infer_b1_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r6c35_infer_b1_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('per call: kernel ms', tot/1e6/35, 'launches', calls/35)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:30]:
    print(f"{int(r['Calls'])/35:6.1f} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6/35:8.3f} ms  {r['Name'][:100]}")
PY
