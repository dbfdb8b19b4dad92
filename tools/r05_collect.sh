#!/bin/bash
# round-5 evidence: bench lines of every workload, family / by-shape tables (eager + hipGraph replay), kernel stats, PMC passes of the dominant kernels,
# the spill table.  Everything lands under gpurun_out/r05_profiles/ (copied into profiles/ by hand).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
RND=r05 bash tools/collect_profiles.sh > $R/gpurun_out/r05_collect.log 2>&1
PMC_ROUND=r05 bash tools/pmc_passes.sh conv_one wgrad_group_one conv_wgrad_one wmsa_one >> $R/gpurun_out/r05_collect.log 2>&1
python3 tools/pmc_summary.py $R/gpurun_out/r05_pmc $R/gpurun_out/r05_profiles/r05_pmc.json >> $R/gpurun_out/r05_collect.log 2>&1
rm -rf $R/gpurun_out/r05_pmc
python3 tools/spill_table.py > $R/gpurun_out/r05_profiles/r05_vgpr_spills_readelf.txt 2>> $R/gpurun_out/r05_collect.log
ls -la $R/gpurun_out/r05_profiles
for f in $R/gpurun_out/r05_profiles/r05_z_bench_line*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['ms_per_step'], d['value'], d['unit'])"; done
