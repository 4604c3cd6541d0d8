# round 5, call 40: race hunt for the side streams at the benchmark's size
cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/stream_race_check.py --reps 4 > gpurun_out/r5c40_race.txt 2> gpurun_out/r5c40_race.err < /dev/null
cat gpurun_out/r5c40_race.txt; tail -3 gpurun_out/r5c40_race.err
