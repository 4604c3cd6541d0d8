# kernel-trace + stats of a short bench run (headline leg only) -> gpurun_out/prof_s ; prints the top kernels per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_s
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s -o s -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg > gpurun_out/prof_s.log 2>&1
rm -f gpurun_out/prof_s/s_kernel_trace.csv
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_s/s_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('kernel ms/step', tot/13/1e6)
for r in rows[:int(__import__('os').environ.get('TOPN','30'))]:
    print(f"{int(r['TotalDurationNs'])/13/1e6:7.3f} ms {int(r['Calls'])/13:6.1f} {float(r['AverageNs'])/1e3:8.1f}us  {r['Name'][:110]}")
P
grep '^{' gpurun_out/prof_s.log | cut -c1-200
