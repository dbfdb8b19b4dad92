#!/bin/bash
# round-4 evidence: bench lines, family / by-shape tables, kernel stats, PMC passes of the three dominant kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
RND=r04 bash tools/collect_profiles.sh > $R/gpurun_out/r04_collect.log 2>&1
PMC_ROUND=r04 bash tools/pmc_passes.sh conv_one wgrad_group_one conv_wgrad_one wmsa_one >> $R/gpurun_out/r04_collect.log 2>&1
python3 tools/pmc_summary.py $R/gpurun_out/r04_pmc $R/gpurun_out/r04_profiles/r04_pmc.json >> $R/gpurun_out/r04_collect.log 2>&1
rm -rf $R/gpurun_out/r04_pmc
ls -la $R/gpurun_out/r04_profiles
for f in $R/gpurun_out/r04_profiles/r04_z_bench_line*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['ms_per_step'], d['value'], d['unit'])"; done
