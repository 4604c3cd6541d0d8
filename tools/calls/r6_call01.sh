# round 6, call 1: the conv weight-gradient stream mismatch -- A/B that names the root cause (amax slots not reserved for the side stream),
# the new batch-8 stream test, the launcher on one GPU over gloo, and this round's starting bench line
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c01
( echo "== A: round-5 behaviour (amax slots NOT reserved), 16-slot pools"; timeout 600 python tools/stream_race_check.py --reps 24 --amax-pool 16 --no-reserve --only-default 2>/dev/null
  echo "== B: slots reserved (the fix), 16-slot pools";                 timeout 600 python tools/stream_race_check.py --reps 32 --amax-pool 16 --only-default 2>/dev/null
  echo "== C: round-5 behaviour, 256-slot pools (the shipped size)";    timeout 600 python tools/stream_race_check.py --reps 32 --no-reserve --only-default 2>/dev/null
  echo "== D: slots reserved, 256-slot pools";                          timeout 600 python tools/stream_race_check.py --reps 32 --only-default 2>/dev/null ) > ${R}_race.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_attention.py -x -q -m gpu -s 2>&1 | tail -15 > ${R}_tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>${R}_bench.err | grep "^{" > ${R}_bench.json
VBG_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>${R}_bench2.err | grep "^{" > ${R}_bench2.json
tail -5 ${R}_bench2.err
cat ${R}_tests.txt; grep -c . ${R}_race.txt; grep "worst over" -A3 ${R}_race.txt; cut -c1-300 ${R}_bench.json
