# round 6, call 4: WHICH intermediate of the 64-channel backward nodes goes wrong first in an outlier run
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c04
for i in 1 2 3; do timeout 500 python tools/stream_race_check.py --reps 40 --only-default --offenders 2e-5 --trace 2>/dev/null | grep -v "noise floor\|e-06 max .* bert" ; done > ${R}_trace.txt 2>&1
grep -c "trace #" ${R}_trace.txt; grep "trace #\|no captured\|lengths differ" ${R}_trace.txt | head -40
