cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-conv3}; K=${2:-fwd}
O=$R/gpurun_out/c3_$W$K
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d ${O}_a -o a --output-format csv -- python $R/tools/conv3_prof.py $W $K > ${O}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d ${O}_b -o b --output-format csv -- python $R/tools/conv3_prof.py $W $K > ${O}_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE -d ${O}_c -o c --output-format csv -- python $R/tools/conv3_prof.py $W $K > ${O}_c.log 2>&1
for f in $(find ${O}_a ${O}_b ${O}_c -name "*counter_collection.csv"); do python $R/tools/pmc_summary.py "gemm_kernel" $f; python $R/tools/pmc_summary.py "conv3x3_kernel" $f; python $R/tools/pmc_summary.py "conv3x3_wgrad_kernel" $f; done 2>&1 | grep -v "dispatches 0" 
