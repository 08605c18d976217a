cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sparse or finishing or heavy or radix or full_path" 2>&1 | tail -15 > gpurun_out/t1.log
timeout 600 python scripts/exp/exp_finish2.py 3000000000 3 > gpurun_out/e1.log 2>&1
cat gpurun_out/t1.log gpurun_out/e1.log
