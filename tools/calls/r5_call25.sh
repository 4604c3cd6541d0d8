#!/bin/bash
# host-side profile of the training step and of single-document inference
cd /root/repo
mkdir -p gpurun_out
python tools/host_profile.py --steps 6 --top 60 > gpurun_out/r5c25_host_train.txt 2> gpurun_out/r5c25_host_train.err
python tools/infer_host_profile.py > gpurun_out/r5c25_host_infer.txt 2> gpurun_out/r5c25_host_infer.err
tail -5 gpurun_out/r5c25_host_train.err
head -3 gpurun_out/r5c25_host_train.txt
