#!/bin/bash
# deferred piece reductions (LAVT_TN_DEFER=0: reduce at once): tests, then A/B bench lines
cd ${GRAFT_REPO_ROOT:-.}
python3 -m pytest tests -m gpu -x -q -k "deferred_piece or ddp or train_step or two_train or full or swin_block" 2>&1 | tail -3
AB_OUT=r06_o_tn_defer_ab.txt AB_CONFIGS=$'defer_off LAVT_TN_DEFER=0\ndefer_on LAVT_TN_DEFER=1' bash tools/r06_ab.sh
