cd ${GRAFT_REPO_ROOT:-.}
python3 -m pytest tests -q -m gpu 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
