cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed"
(timeout 600 $B --k 21; timeout 600 $B --mode genome --genome-len 7500000000; timeout 600 $B --mode genome --genome-len 2500000000;  timeout 600 $B ) 2>&1 | grep -v amdgpu > gpurun_out/bench_k21_cov1.log
python - <<'PY'
import json
for line in open("gpurun_out/bench_k21_cov1.log"):
    if line.startswith("{"):
        d = json.loads(line)
        print(d["config"].get("workload"), d["value"], d["ms_per_step"], d.get("parity_fullsize"), {k: round(v, 1) for k, v in d.get("kernels_ms", d.get("stage_ms", {})).items()} if isinstance(d.get("kernels_ms", d.get("stage_ms")), dict) else "")
    else:
        print(line.rstrip()[:300])
PY
