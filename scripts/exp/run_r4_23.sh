cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
(time timeout 900 python bench.py) > gpurun_out/bench_default.log 2>&1
python - <<'PY'
import json
for line in open("gpurun_out/bench_default.log"):
    if line.startswith("{"):
        d = json.loads(line)
        print(d["value"], d["ms_per_step"], d["steps"], d["warmup"], d["parity_fullsize"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"], d.get("extra_keys"), d["extra"]["seconds"])
        print({k: (v.get("ms_per_step"), v.get("parity")) for k, v in d["extra"].items() if isinstance(v, dict)})
    elif line.startswith("real"): print(line.strip())
PY
