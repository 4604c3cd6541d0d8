# round 6, call 3: is the intermittent mismatch a STREAM race at all?  (a) NaN poison in every fresh torch.empty of the package, site by
# site, on ONE stream; (b) one stream with a shifted allocation pattern every step
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c03
timeout 1500 python tools/poison_check.py > ${R}_poison.txt 2>${R}_poison.err
timeout 600 python tools/stream_race_check.py --reps 40 --jitter --offenders 2e-5 2>/dev/null | grep -v "noise floor" > ${R}_jitter.txt
tail -5 ${R}_poison.err; grep -c . ${R}_poison.txt; grep "POISON\|sites" ${R}_poison.txt | cut -c1-400 | head -40
grep -c offender ${R}_jitter.txt; grep "worst over" -A3 ${R}_jitter.txt
