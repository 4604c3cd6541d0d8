# round 3, GPU call 1: pair-GEMM probe + TCC hit/miss counters of the existing bf16x3 plane kernel
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 600 $R/tools/probes/pair_gemm_probe > $R/gpurun_out/pair_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for T in 128129 256128; do
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pg_pmc_c_$T -o c --output-format csv -- python $R/tools/plane_gemm_prof.py 4128 3072 768 $T 20 > $R/gpurun_out/pg_pmc_c_$T.log 2>&1
python $R/tools/pmc_summary.py plane_gemm $(find $R/gpurun_out/pg_pmc_c_$T -name "*counter_collection.csv") > $R/gpurun_out/pg_pmc_c_$T.txt 2>&1
done
cat $R/gpurun_out/pair_probe.txt | tail -120
cat $R/gpurun_out/pg_pmc_c_*.txt
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/call1_pytest.txt
cat gpurun_out/call1_pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/call1_bench.json 2> gpurun_out/call1_bench.err
tail -c 3000 gpurun_out/call1_bench.json
