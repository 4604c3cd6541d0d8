cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-128128}
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pg_pmc_c_$T -o c --output-format csv -- python $R/tools/plane_gemm_prof.py 4128 3072 768 $T 20 > $R/gpurun_out/pg_pmc_c_$T.log 2>&1
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pg_pmc_d_$T -o d --output-format csv -- python $R/tools/plane_gemm_prof.py 4128 3072 768 $T 20 > $R/gpurun_out/pg_pmc_d_$T.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES -d $R/gpurun_out/pg_pmc_e_$T -o e --output-format csv -- python $R/tools/plane_gemm_prof.py 4128 3072 768 $T 20 > $R/gpurun_out/pg_pmc_e_$T.log 2>&1
tail -2 $R/gpurun_out/pg_pmc_c_$T.log $R/gpurun_out/pg_pmc_d_$T.log $R/gpurun_out/pg_pmc_e_$T.log
