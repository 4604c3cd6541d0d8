# round 6, call 15: FPN lateral convolutions on a stream of their own (VBG_FPN_STREAM): A/B x 3, race check (heads stream now default on)
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c15
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config']['last_loss'])" | tee -a ${R}_ab.txt; }
rm -f ${R}_ab.txt
for i in 1 2 3; do run VBG_FPN_STREAM=0; run VBG_FPN_STREAM=1; done
VBG_FPN_STREAM=1 timeout 600 python tools/stream_race_check.py --reps 60 --only-default --offenders 2e-5 2>/dev/null | grep -v "noise floor #" | tail -4 | tee ${R}_race.txt
