#!/bin/bash
# second evidence run of round 6 (after the attention kernels' instruction diet and the deferred piece reductions): tools/r06_final.sh, the 1-rank-group record,
# a PMC pass of the 392-token attention kernels
cd ${GRAFT_REPO_ROOT:-.}
bash tools/r06_final.sh
bash tools/r06_rank1.sh > /dev/null 2>&1
ATTN_WD=8 ATTN_WS=7 ATTN_NWIN=16 PMC_ROUND=r06b bash tools/pmc_passes.sh attn_one > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/r06b_pmc gpurun_out/r06_pmc_attn_392_tokens_after_diet.json > /dev/null 2>&1
rm -rf gpurun_out/r06b_pmc
cat gpurun_out/r06_d_one_rank_group_buckets.txt
python3 -c "
import json; d=json.load(open('gpurun_out/r06_pmc_attn_392_tokens_after_diet.json'))
for k,v in d.items(): print(k, json.dumps(v.get('derived'))[:700])"
