# round 4, call 5: where the generic GEMM launches and the row-reuse convolution launches of a step go (per shape, event time)
cd /root/repo
python tools/step_gemm_profile.py > gpurun_out/r4c5_step_gemm.txt 2>/dev/null; cat gpurun_out/r4c5_step_gemm.txt | head -50
python tools/step_conv3_profile.py > gpurun_out/r4c5_step_conv3.txt 2>/dev/null; cat gpurun_out/r4c5_step_conv3.txt | head -40
