cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/exp/exp_finish2.py 3000000000 3 parity 2 > gpurun_out/e6_default.log 2>&1
for v in wg3p wg1u4 wg3u4 wg5u1 b12; do
  timeout 300 python scripts/bin/$v/scripts/exp/exp_finish2.py 3000000000 3 noparity 2 > gpurun_out/e6_$v.log 2>&1
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sparse or finishing or heavy or radix or full_path" 2>&1 | tail -15 > gpurun_out/t6.log
tail -n 3 gpurun_out/e6_*.log gpurun_out/t6.log
