#!/bin/bash
# final validation + evidence of round 5: the whole GPU test-suite, the smoke entry, then tools/r05_collect.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/r05_collect.sh 2>&1 | tail -12
