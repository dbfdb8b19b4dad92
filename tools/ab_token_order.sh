for v in 1 0 1 0; do LAVT_TOKEN_ORDER_WGRAD=$v python bench.py --steps 100 --warmup 10 > /tmp/b_$v.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("/tmp/b_$v.json").read().strip().splitlines()[-1]); print("tok $v", d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel"))
PY
done
