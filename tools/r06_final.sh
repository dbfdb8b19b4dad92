#!/bin/bash
# validation + evidence of round 6: the whole GPU test-suite, the smoke entry, then tools/collect_profiles.sh (bench lines of every workload, kernel stats, by-shape tables, step timeline)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r06_profiles
python3 -m pytest tests -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r06_profiles/r06_z_gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r06_profiles/r06_z_gpu_tests.txt
RND=r06 bash tools/collect_profiles.sh > gpurun_out/r06_profiles/collect.log 2>&1
for f in gpurun_out/r06_profiles/r06_z_bench_line*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['ms_per_step'], d['value'], d['unit'], d.get('roofline',{}).get('frac'))"; done
