cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
nproc
for t in 8 16 32 64; do
  BNPK_READ_THREADS=$t timeout 600 python scripts/exp/exp_reference_loop.py 8000000 31 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print($t, 'stream', d['stream_form']['chunk_268435456']['gbases_per_s'], d['stream_form']['chunk_5000000']['gbases_per_s'], 'example', d['example_form']['chunk_5000000']['gbases_per_s'], d['example_form']['chunk_268435456']['gbases_per_s'])"
done
