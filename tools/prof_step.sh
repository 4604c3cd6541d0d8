# kernel-trace + stats of a short bench run (headline leg only) -> gpurun_out/prof_s ; prints the top kernels per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_s
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s -o s -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg > gpurun_out/prof_s.log 2>&1
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_s/s_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('kernel ms/step', tot/23/1e6)
for r in rows[:int(__import__('os').environ.get('TOPN','30'))]:
    print(f"{int(r['TotalDurationNs'])/23/1e6:7.3f} ms {int(r['Calls'])/23:6.1f} {float(r['AverageNs'])/1e3:8.1f}us  {r['Name'][:110]}")
P
python - <<'P'
import csv
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in csv.DictReader(open('gpurun_out/prof_s/s_kernel_trace.csv'))]
rows.sort()
# the timed region = the last 10 of 13 steps: take the last 10/13 of the launches
n=len(rows); per=n//23; seg=rows[n-10*per:]
busy=sum(e-s for s,e in seg); span=seg[-1][1]-seg[0][0]
gaps=[seg[i+1][0]-seg[i][1] for i in range(len(seg)-1)]
big=sorted(gaps)[-10:]
print(f"launches/step {per}  span/step {span/10/1e6:.2f} ms  busy/step {busy/10/1e6:.2f} ms  idle {(span-busy)/10/1e6:.2f} ms; gaps >20us: {sum(1 for g in gaps if g>20000)/10:.0f}/step totalling {sum(g for g in gaps if g>20000)/10/1e6:.2f} ms; largest {[round(g/1e3) for g in big]} us")
P
rm -f gpurun_out/prof_s/s_kernel_trace.csv
grep '^{' gpurun_out/prof_s.log | cut -c1-200
