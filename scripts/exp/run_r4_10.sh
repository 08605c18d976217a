cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_reader_big_batches.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/t_reader.log
cat gpurun_out/t_reader.log
timeout 300 python scripts/exp/exp_small_chunks.py 2>&1 | grep -v amdgpu | head -12
