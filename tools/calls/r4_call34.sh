# round 4, call 34: the whole GPU suite + smoke on the final build
cd /root/repo
timeout 2700 python -m pytest tests -q -m gpu --tb=short -rf 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
