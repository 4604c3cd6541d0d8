# round 6, call 50: the committed profiles, re-collected on the final tree
cd /root/repo
mkdir -p gpurun_out
bash tools/collect_profiles.sh > gpurun_out/r6c50_collect.log 2>&1
tail -5 gpurun_out/r6c50_collect.log
