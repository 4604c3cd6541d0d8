cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_g
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_g -o g -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg > gpurun_out/prof_g.log 2>&1
tail -2 gpurun_out/prof_g.log | cut -c1-300
python tools/gap_report.py gpurun_out/prof_g/g_kernel_trace.csv 23 30 10
rm -f gpurun_out/prof_g/g_kernel_trace.csv
