cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_collectives.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/t_coll.log
cat gpurun_out/t_coll.log
