# usage: ab_lib.sh <other liblavt_hip.so> [env for that arm]  -- alternates the in-tree build and another build on one box
for arm in new old new old; do
  if [ $arm = old ]; then env LAVT_LIB_PATH=$1 $2 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > /tmp/b_$arm.json 2>/dev/null
  else python bench.py --steps 100 --warmup 10 --no-cpu-baseline > /tmp/b_$arm.json 2>/dev/null; fi
  python - <<PY
import json
d=json.loads(open("/tmp/b_$arm.json").read().strip().splitlines()[-1]); print("$arm", d["ms_per_step"], d["value"])
PY
done
