# round 6, call 13: the whole -m gpu suite (no -x) + smoke
cd /root/repo
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r6c13_pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -1 >> gpurun_out/r6c13_pytest.txt
grep -v Warning gpurun_out/r6c13_pytest.txt | tail -12
