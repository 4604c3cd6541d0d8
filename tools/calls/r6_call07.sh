# round 6, call 7: the guard left out (VBG_DEBUG_CONV3_NO_RING_GUARD=1) brings the outliers back -- step-level proof, and the pytest gate sees it
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c07
( echo "== guard OUT (VBG_DEBUG_CONV3_NO_RING_GUARD=1)"
  for i in 1 2; do VBG_DEBUG_CONV3_NO_RING_GUARD=1 timeout 600 python tools/stream_race_check.py --reps 80 --only-default --trace --offenders 1 2>/dev/null | grep "conv weight\|trace #" | grep -v "cnn: rel-L2 [0-9.]*e-06" ; done
  echo "== guard IN"
  for i in 1 2; do timeout 600 python tools/stream_race_check.py --reps 80 --only-default --trace --offenders 1 2>/dev/null | grep "conv weight\|trace #" | grep -v "cnn: rel-L2 [0-9.]*e-06" ; done
  echo "== pytest gate with the guard OUT (expected to FAIL)"
  VBG_DEBUG_CONV3_NO_RING_GUARD=1 timeout 900 python -m pytest tests/test_gpu_streams.py -x -q -m gpu -k default_streams 2>&1 | grep "passed\|failed\|AssertionError: run" | cut -c1-300
  echo "== pytest gate with the guard IN"
  timeout 900 python -m pytest tests/test_gpu_streams.py -x -q -m gpu 2>&1 | grep "passed\|failed" ) > ${R}_ab.txt 2>&1
cut -c1-260 ${R}_ab.txt
