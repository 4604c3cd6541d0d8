cd /tmp && export TMPDIR=/tmp
VBG_INFER_BATCHES=8 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_inf -o inf -- python /root/repo/tools/infer_latency.py > /tmp/inf.log 2>&1
f=$(find /tmp/prof_inf -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print(list(rows[0].keys()))
# find the last full call: split at embed_ln_fwd_kernel? use normalize_resize_kernel as the first kernel of a call
idx=[i for i,r in enumerate(rows) if 'embed_ln_fwd' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
t0=int(rows[a]['Start_Timestamp'])
out=open('/root/repo/gpurun_out/r6c54_infer_b8_trace.txt','w')
prev_end=t0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    g=(int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])))*(int(r['Grid_Size_Y'])//max(1,int(r['Workgroup_Size_Y'])))*(int(r['Grid_Size_Z'])//max(1,int(r['Workgroup_Size_Z'])))
    out.write(f"{(s-t0)/1e3:9.1f} us  gap {(s-prev_end)/1e3:6.1f}  dur {(e-s)/1e3:7.1f} us  blocks {g:6d} x {r['Workgroup_Size_X']:>4s}  {r['Kernel_Name'][:90]}\n")
    prev_end=max(prev_end,e)
out.write(f"call: {(prev_end-t0)/1e3:.1f} us wall, {b-a} launches\n")
PY
tail -1 /root/repo/gpurun_out/r6c54_infer_b8_trace.txt
