# round 4, call 17: the whole -m gpu suite + smoke on the current build
cd /root/repo
python -m pytest tests -q -m gpu > gpurun_out/r4c17_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4c17_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
