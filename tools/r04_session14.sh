#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s14
mkdir -p $O
cd $R
timeout 600 python3 bench.py --steps 60 --no-cpu-baseline --no-profile --workload video_swin_b_t8_384 > $O/bench_video.json 2> $O/bench_video.err; echo "video rc=$?" >> $O/summary.txt
LAVT_PROBE=0,0,0,1 timeout 600 python3 bench.py --steps 60 --no-cpu-baseline --no-profile --workload video_swin_b_t8_384 > $O/bench_video_riders.json 2> $O/bench_video_riders.err; echo "video riders rc=$?" >> $O/summary.txt
LAVT_DTABLE_CHAIN=0 timeout 600 python3 bench.py --steps 60 --no-cpu-baseline --no-profile --workload video_swin_b_t8_384 > $O/bench_video_nochain.json 2> $O/bench_video_nochain.err; echo "video nochain rc=$?" >> $O/summary.txt
cat $O/summary.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s14/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
