#!/bin/bash
# whole GPU test-suite + smoke + a bench line of the current tree
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/r05_full_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O/r05_full_tests.txt
python bench.py --no-cpu-baseline > $O/r05_b_bench.json 2> $O/r05_b_bench.err
cat $O/r05_full_tests.txt; python -c "
import json; d=json.loads(open('$O/r05_b_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel'][:80], d['roofline']['frac'], d['roofline'].get('avg_launch_us'), d['config']['profile']['scope_us_per_step'], d['config'].get('wmsa_pwam_mfma_frac'))"
