#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r05_pieces2.txt; rm -f $T
python -m pytest tests -q -m gpu -k "tn_grouped or token_order or rider or partial_tiles" 2>&1 | tail -3 >> $T
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/tools/ln_rider_time.py > /dev/null 2>&1
f=$(ls $O/tr/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-200 >> $T; rm -rf $O/tr
cat $T
