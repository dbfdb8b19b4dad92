#!/bin/bash
# the N > 1 code path (SyncBN exchanges + bucketed all-reduces inside the captured step) in a 1-rank RCCL group on the 1-GPU box, against the single-process step;
# then the 1-rank-group harness test at three bucket sizes (bucket readiness depends on the order of the grouped launches, which round 6 changed)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r06_d_one_rank_group_buckets.txt; rm -f $T
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['config'].get('rccl'))" >> $T; }
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | line "single process"
for mib in 16 32 64; do
  LAVT_FORCE_COLLECTIVES=1 LAVT_BUCKET_MIB=$mib python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | line "1-rank group fp32 buckets ${mib}MiB"
done
LAVT_FORCE_COLLECTIVES=1 LAVT_BUCKET_MIB=32 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-optimizer --bf16-buckets 2>/dev/null | line "1-rank group bf16 buckets 32MiB"
for mib in 16 32 64; do
  echo "test_ddp_step_graph_equals_eager_in_one_rank_group at LAVT_BUCKET_MIB=$mib: $(LAVT_BUCKET_MIB=$mib timeout 600 python3 -m pytest tests/test_gpu_modules.py -q -k 'one_rank_group' 2>&1 | tail -1)" >> $T
done
cat $T
