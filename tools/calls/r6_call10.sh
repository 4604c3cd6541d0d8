# round 6, call 10: is the chip ever idle in the three-stream step?  kernel trace of the headline leg -> union of the kernel intervals
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_u
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_u -o u -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass > gpurun_out/prof_u.log 2>&1
# (the roofline leg re-runs the 10 steps with event pairs: 23 steps in the trace, the LAST 10 instrumented; take the 10 before them)
python tools/busy_union.py gpurun_out/prof_u/u_kernel_trace.csv 23 20 | tee gpurun_out/r6c10_union.txt
rm -f gpurun_out/prof_u/u_kernel_trace.csv
grep '^{' gpurun_out/prof_u.log | cut -c1-160
