#!/bin/bash
# round 5: the pipelined grouped weight gradient (csrc/gemm_tn_pipe.hip) -- parity tests, the launch alone, the step with / without it
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out
python -m pytest tests -q -m gpu -k "tn or grouped or wgrad or rider or train_step or ddp or adamw or bench_step" 2>&1 | tail -12 > $O/r05_tnp_tests.txt
for pipe in 0 1; do for st in 4 3; do
  [ $pipe = 0 ] && [ $st = 3 ] && continue
  echo "LAVT_TN_PIPE=$pipe stages=$st" >> $O/r05_tnp_time.txt
  LAVT_TN_PIPE=$pipe LAVT_TN_PIPE_STAGES=$st python tools/ln_rider_time.py >> $O/r05_tnp_time.txt 2>&1
done; done
for v in 1 0 1 0; do LAVT_TN_PIPE=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile > /tmp/b_$v.json 2>/dev/null; python - <<PY >> $O/r05_tnp_time.txt
import json
d=json.loads(open("/tmp/b_$v.json").read().strip().splitlines()[-1]); print("LAVT_TN_PIPE=$v", d["ms_per_step"], d["value"], d["config"].get("loss_check"))
PY
done
LAVT_TN_PIPE_STAGES=3 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stages=3', d['ms_per_step'])" >> $O/r05_tnp_time.txt
cat $O/r05_tnp_tests.txt $O/r05_tnp_time.txt
