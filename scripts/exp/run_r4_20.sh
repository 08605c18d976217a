cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_api.py tests/test_gpu_parity.py tests/test_errors.py tests/test_reader_big_batches.py -x -q -m gpu 2>&1 | tail -4
timeout 900 python scripts/bench_configs.py 50000000 2> gpurun_out/configs.err | tail -1 > gpurun_out/configs.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/configs.json"))
for k, v in d.items():
    print(k, {a: b for a, b in v.items() if a in ("ms_per_step", "kernels_ms", "kernel_ms", "seconds", "gbases_per_s", "build_s", "lookup_s")} if isinstance(v, dict) else v)
PY
