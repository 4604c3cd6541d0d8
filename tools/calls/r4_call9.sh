# round 4, call 9: the whole -m gpu suite on the current build
cd /root/repo
python -m pytest tests -q -m gpu -x > gpurun_out/r4c9_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4c9_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
