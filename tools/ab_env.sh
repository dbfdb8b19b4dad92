# usage: ab_env.sh VAR  -- bench with VAR=1 / VAR=0 alternating (same box)
for v in 1 0 1 0; do env $1=$v python bench.py --steps 100 --warmup 10 > /tmp/b_$v.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("/tmp/b_$v.json").read().strip().splitlines()[-1]); print("$1=$v", d["ms_per_step"], d["value"])
PY
done
