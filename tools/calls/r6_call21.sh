# round 6, call 21: the default streams (incl. the heads' stream) against one stream on the OTHER shapes and under autocast
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c21
( for a in "--amp" "--shape cfg3" "--shape cfg4" "--shape cfg5 --reps 12" "--shape cfg4 --amp"; do echo "== $a"; timeout 900 python tools/stream_race_check.py --reps 24 --only-default --offenders 2e-5 $a 2>/dev/null | grep -v "noise floor #" | grep "offender\|worst over" -A4 | grep -v "^--" | tail -6; done ) > ${R}_race.txt 2>&1
cat ${R}_race.txt | cut -c1-200
