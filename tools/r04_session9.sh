#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s9
mkdir -p $O
cd $R
timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
LAVT_DTABLE_CHAIN=0 timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench_nochain.json 2> $O/bench_nochain.err; echo "nochain rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench2.json 2> $O/bench2.err; echo "bench2 rc=$?" >> $O/summary.txt
timeout 2400 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_full.py -q -m gpu > $O/pytest_mod.log 2>&1; echo "mod rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 50 --no-cpu-baseline --no-profile --workload video_swin_b_t8_384 > $O/bench_video.json 2> $O/bench_video.err; echo "video rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-profile --workload swin_t_w7_480_b8 > $O/bench_swint.json 2> $O/bench_swint.err; echo "swint rc=$?" >> $O/summary.txt
tail -n 6 $O/pytest_mod.log; cat $O/summary.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s9/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'], d['config'].get('loss_check'))
    except Exception as e: print(f, 'ERR', e)
PY
