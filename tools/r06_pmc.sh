#!/bin/bash
# round-6 PMC passes (single counter per run, tools/pmc_passes.sh) of the dominant kernels -> gpurun_out/r06_pmc.json
cd ${GRAFT_REPO_ROOT:-.}
PMC_ROUND=r06 bash tools/pmc_passes.sh wgrad_group_one conv_one wmsa_one attn_one > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/r06_pmc gpurun_out/r06_pmc.json
rm -rf gpurun_out/r06_pmc
python3 -c "
import json; d=json.load(open('gpurun_out/r06_pmc.json'))
for k,v in d.items(): print(k, json.dumps(v.get('derived'))[:600])"
