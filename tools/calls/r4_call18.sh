# round 4, call 18: reduction order (chunk, kh, kw) of the row-reuse forward kernels: tests, traffic of the 256 -> 256 launch at 128^2, shapes, step
cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/c3_fetch -o f --output-format csv -- python $R/tools/conv3_prof.py pw fwd > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/c3_write -o w --output-format csv -- python $R/tools/conv3_prof.py pw fwd > /dev/null 2>&1
cd $R
for f in gpurun_out/c3_fetch/f_counter_collection.csv gpurun_out/c3_write/w_counter_collection.csv; do python tools/pmc_summary.py "conv3x3_kernel" $f; done
python tools/conv3_pw_bench.py 2>&1 | grep -E "PW  " | grep -v forced | head -16
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_us'])"
