#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for st in 4 3; do echo "LAVT_TN_PIPE_STAGES=$st"; LAVT_TN_PIPE_STAGES=$st timeout 300 python3 tools/probes/overlap_probe.py 2>&1 | tail -6; done
echo "LAVT_TN_PIPE=0"; LAVT_TN_PIPE=0 timeout 300 python3 tools/probes/overlap_probe.py 2>&1 | tail -6
