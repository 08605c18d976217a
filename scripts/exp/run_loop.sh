# the reader / API tests and the chunk loop's rates: gpurun --timeout 1800 -- 'bash scripts/exp/run_loop.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_reader_big_batches.py tests/test_api.py tests/test_gzip_reading.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python scripts/exp/exp_reference_loop.py 8000000 31 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/reference_loop.json
python -c "
import json; d=json.load(open('gpurun_out/reference_loop.json'))
for k in ('example_form','library_form','stream_form'): print(k, {a:(b['ms'], b['gbases_per_s'], b['same_histogram']) if isinstance(b, dict) else b for a,b in d[k].items()})"
