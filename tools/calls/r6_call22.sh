# round 6, call 22: kernel trace of the headline leg kept whole (queue ids): who is alone on the chip, and when
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_q
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_q -o q -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass > gpurun_out/prof_q.log 2>&1
head -2 gpurun_out/prof_q/q_kernel_trace.csv | cut -c1-400
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_q/q_kernel_trace.csv')))
# keep only the 6 timed headline steps: steps 3..8 of 15 (3 warm-up + 6 + 6 roofline leg)
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows); per=n//15
seg=rows[3*per:9*per]
import gzip, json
out=[(int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id','?'), r['Kernel_Name'][:90]) for r in seg]
json.dump(out, gzip.open('gpurun_out/r6c22_trace.json.gz','wt'))
print(len(out), 'launches kept;', len(set(o[2] for o in out)), 'queues')
P
rm -f gpurun_out/prof_q/q_kernel_trace.csv
