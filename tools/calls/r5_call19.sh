# round 5, call 19: the whole GPU suite and smoke on the current tree
cd /root/repo
( time timeout 3300 python -m pytest tests -q -m gpu --tb=short -rf 2>&1 | tail -15 ) 2>&1 | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
