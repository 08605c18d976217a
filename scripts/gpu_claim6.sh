cd $GRAFT_REPO_ROOT
R=gpurun_out/r05claim
mkdir -p $R
timeout 1200 python -m pytest tests -m gpu -q > $R/pytest.log 2>&1; tail -4 $R/pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed --no-extra"
for cfg in "--mode genome" "--k 21" "--mode genome --genome-len 2500000000" "--mode genome --genome-len 7500000000" "--k 15" "--k 27 --canonical"; do
  timeout 300 $B $cfg > $R/b.json 2>> $R/bench.err; python - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r05claim/b.json"))
print(sys.argv[1], "|", d["value"], "Gbases/s", d["ms_per_step"], "ms parity", d["parity_fullsize"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if v["ms_per_step"]>1})
PY
done
