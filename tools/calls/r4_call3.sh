# round 4, call 3: counters of the PW convolution kernel (256 -> 256 at 128^2, and the one-round 256 @ 32^2 shape); stock-DDP test again
cd /root/repo
python -m pytest tests/test_gpu_ddp.py -x -q -m gpu -s -k stock > gpurun_out/r4c3_ddp.txt 2>&1; echo "ddp rc=$?"; grep -n "losses stock\|running mean\|rel-L2 of the\|passed\|failed" gpurun_out/r4c3_ddp.txt | cut -c1-600
bash tools/prof_conv3.sh pw fwd > gpurun_out/r4c3_pw_pmc.txt 2>&1; cat gpurun_out/r4c3_pw_pmc.txt | grep -v "^$" | head -60
