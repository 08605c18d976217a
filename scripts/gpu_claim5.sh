cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05claim
timeout 900 python -m pytest tests/test_finish_modes.py -x -q 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "count_sparse or radix" 2>&1 | tail -5
timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-host-fed --cpu-sample-reads 50000 > gpurun_out/r05claim/bench.json 2> gpurun_out/r05claim/bench.err; cut -c1-300 gpurun_out/r05claim/bench.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05claim/bench.json"))
print(d["value"], d["ms_per_step"], d["parity_fullsize"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})
PY
tail -3 gpurun_out/r05claim/bench.err
