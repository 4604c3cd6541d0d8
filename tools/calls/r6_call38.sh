cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "small_tile or pair_form or one_product" 2>&1 | tail -4 | tee gpurun_out/r6c38_pytest.txt
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_layer_entry.py tests/test_gpu_full_scale.py tests/test_gpu_attention.py -x -q 2>&1 | tail -6 | tee -a gpurun_out/r6c38_pytest.txt
for v in 0 1; do echo "VBG_PAIR_SMALL=$v"; VBG_PAIR_SMALL=$v timeout 300 python tools/infer_latency.py 2>/dev/null; done | tee gpurun_out/r6c38_infer.txt
cd /tmp && export TMPDIR=/tmp
VBG_INFER_BATCHES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o inf -- python /root/repo/tools/infer_latency.py > /tmp/inf.log 2>&1
f=$(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1); cp "$f" /root/repo/gpurun_out/r6c38_infer_b1_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r6c38_infer_b1_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('per call: kernel ms', tot/1e6/35, 'launches', calls/35)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print(f"{int(r['Calls'])/35:6.1f} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6/35:8.3f} ms  {r['Name'][:100]}")
PY
