# round 6, call 8: the whole -m gpu suite on the tree with the ring fix, the launcher, the shared-communicator default and full_cfg3e8
cd /root/repo
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6c08_pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -2 >> gpurun_out/r6c08_pytest.txt
cat gpurun_out/r6c08_pytest.txt
