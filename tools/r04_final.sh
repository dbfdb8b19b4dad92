#!/bin/bash
# final validation + evidence of a round: the whole GPU test-suite, the smoke entry, then tools/r04_collect.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/r04_collect.sh 2>&1 | tail -12
