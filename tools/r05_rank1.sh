#!/bin/bash
# the N > 1 code path (SyncBN exchanges + bucketed all-reduces inside the captured step) in a 1-rank RCCL group on the 1-GPU box, against the single-process step
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r05_rank1.txt; rm -f $T
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['config'].get('rccl'))" >> $T; }
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | line "single process"
for mib in 32 64 128; do
  LAVT_FORCE_COLLECTIVES=1 LAVT_BUCKET_MIB=$mib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | line "1-rank group fp32 buckets ${mib}MiB"
done
LAVT_FORCE_COLLECTIVES=1 LAVT_BUCKET_MIB=64 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-optimizer --bf16-buckets 2>/dev/null | line "1-rank group bf16 buckets 64MiB"
LAVT_FORCE_COLLECTIVES=1 LAVT_BUCKET_MIB=64 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-optimizer --rccl-channels 4 2>/dev/null | line "1-rank group fp32 64MiB, 4 channels"
cat $T
