#!/bin/bash
# One-call multi-GPU sweep for the day an N-GPU MI355X node is leased (every run a FRESH `python bench.py --gpus N` child: bench.py starts and
# supervises its own ranks, 127.0.0.1 rendezvous):
#     N in {1,2,4,8} (capped at the GPUs present)  x  gradient buckets {fp32, bf16}  x  --rccl-channels {default,4,8,16}
# JSON lines are collected under ${OUT:-gpurun_out/scale_sweep}/, one file per cell, plus scale_sweep_summary.txt (ms/step, images/s, scaling vs N=1
# of the same bucket dtype / channel setting).  STEPS / WARMUP / WORKLOAD / CHANNELS / NS override the grid.
#     bash tools/scale_sweep.sh                 # full grid
#     NS="1 8" CHANNELS="0 8" bash tools/scale_sweep.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=${OUT:-$R/gpurun_out/scale_sweep}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
HAVE=$(python3 -c 'import torch; print(torch.cuda.device_count())')
NS=${NS:-"1 2 4 8"}
CHANNELS=${CHANNELS:-"0 4 8 16"}          # 0 = RCCL's default channel count
STEPS=${STEPS:-100}
WARMUP=${WARMUP:-10}
WORKLOAD=${WORKLOAD:-swin_b_w12_480_b2}
echo "# $(date -u +%FT%TZ) GPUs present: $HAVE; workload $WORKLOAD; steps $STEPS warmup $WARMUP" > "$OUT/scale_sweep_summary.txt"
for n in $NS; do
  [ "$n" -gt "$HAVE" ] && { echo "# N=$n skipped: only $HAVE GPUs" >> "$OUT/scale_sweep_summary.txt"; continue; }
  for bk in fp32 bf16; do
    for ch in $CHANNELS; do
      [ "$n" -eq 1 ] && { [ "$bk" = bf16 ] || [ "$ch" != 0 ]; } && continue          # one GPU: no collectives, one cell
      tag="n${n}_${bk}_ch${ch}"
      args="--gpus $n --steps $STEPS --warmup $WARMUP --workload $WORKLOAD --no-cpu-baseline --no-profile"
      [ "$bk" = bf16 ] && args="$args --bf16-buckets"
      [ "$ch" != 0 ] && args="$args --rccl-channels $ch"
      # shellcheck disable=SC2086
      timeout ${CELL_TIMEOUT_S:-900} python3 bench.py $args > "$OUT/$tag.json" 2> "$OUT/$tag.err"
      rc=$?
      python3 - "$OUT/$tag.json" "$tag" "$rc" >> "$OUT/scale_sweep_summary.txt" <<'PY'
import json, sys
fn, tag, rc = sys.argv[1:4]
try:
    d = json.loads(open(fn).read().strip().splitlines()[-1])
    print(f"{tag:18s} rc={rc} n_gpus={d['n_gpus']} ms/step={d['ms_per_step']:.3f} images/s={d['value']:.1f} graph={d['config'].get('hip_graph')} rccl={d['config'].get('rccl')}")
except Exception as e:          # noqa: BLE001
    print(f"{tag:18s} rc={rc} no JSON line ({type(e).__name__}); see {fn[:-5]}.err")
PY
    done
  done
done
python3 - "$OUT" >> "$OUT/scale_sweep_summary.txt" <<'PY'
import glob, json, os, sys
rows = {}
for fn in glob.glob(os.path.join(sys.argv[1], "n*_*.json")):
    try:
        d = json.loads(open(fn).read().strip().splitlines()[-1])
        rows[os.path.basename(fn)[:-5]] = d
    except Exception:          # noqa: BLE001
        pass
base = rows.get("n1_fp32_ch0")
if base:
    print("# scaling efficiency = value(N) / (N * value(1)); weak scaling, 1-GPU line = n1_fp32_ch0")
    for tag in sorted(rows, key=lambda t: (int(t.split('_')[0][1:]), t)):
        d = rows[tag]
        print(f"{tag:18s} speed-up {d['value'] / base['value']:.2f}x  efficiency {d['value'] / (d['n_gpus'] * base['value']):.3f}")
PY
cat "$OUT/scale_sweep_summary.txt"
