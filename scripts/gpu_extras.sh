#!/bin/bash
# one GPU call: the default bench with its extras, then the whole -m gpu suite
mkdir -p gpurun_out
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_extras.json 2> gpurun_out/bench_extras.err
echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_extras.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k, v in d.get("extra", {}).items():
    if isinstance(v, dict):
        print(k, {x: v[x] for x in v if x in ("ms_per_step", "parity", "error", "read_gb_per_s", "decode_gb_per_s", "build_ms", "flags_left_the_device", "gb_per_s")})
PY
tail -5 gpurun_out/bench_extras.err
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
