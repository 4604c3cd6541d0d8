# round 5, call 22: the whole GPU suite + smoke on the final tree
cd /root/repo
( time timeout 3300 python -m pytest tests -q -m gpu --tb=short -rf 2>&1 | grep -v "Warning\|warn\|^  \|^$\|tests/test_" | tail -12 ) 2>&1 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
