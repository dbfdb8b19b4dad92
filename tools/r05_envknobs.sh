#!/bin/bash
# runtime knobs of the HIP runtime against the captured step (same box, back to back)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['loss_check']['ok'])"; }
echo "default: $(run)"
echo "HIP_FORCE_DEV_KERNARG=1: $(HIP_FORCE_DEV_KERNARG=1 run)"
echo "HIP_FORCE_DEV_KERNARG=0: $(HIP_FORCE_DEV_KERNARG=0 run)"
echo "AMD_OPT_FLUSH=0: $(AMD_OPT_FLUSH=0 run)"
echo "AMD_OPT_FLUSH=3: $(AMD_OPT_FLUSH=3 run)"
echo "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: $(DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run)"
echo "DEBUG_HIP_GRAPH_BATCH_SIZE=1024: $(DEBUG_HIP_GRAPH_BATCH_SIZE=1024 run)"
echo "GPU_MAX_HW_QUEUES=1: $(GPU_MAX_HW_QUEUES=1 run)"
echo "default: $(run)"
