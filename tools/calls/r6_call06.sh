# round 6, call 6: can the kernel-level contention test SEE the ring race?  (guard out: expect mismatches; guard in: expect 0)
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c06
( VBG_DEBUG_CONV3_NO_RING_GUARD=1 timeout 600 python tools/ring_guard_ab.py 3000 2>&1 | tail -1
  timeout 600 python tools/ring_guard_ab.py 3000 2>&1 | tail -1
  VBG_DEBUG_CONV3_NO_RING_GUARD=1 timeout 600 python tools/stream_race_check.py --reps 80 --only-default 2>/dev/null | grep "worst over" -A2 | grep cnn ) > ${R}_ab.txt 2>&1
cat ${R}_ab.txt
timeout 900 python -m pytest tests/test_gpu_streams.py -x -q -m gpu 2>&1 | tail -3
