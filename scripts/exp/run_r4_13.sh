cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_reader_big_batches.py tests/test_api.py tests/test_gpu_parity.py tests/test_errors.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/t_reader.log
cat gpurun_out/t_reader.log
timeout 600 python scripts/exp/exp_reference_loop.py 8000000 31 2>&1 | grep -v amdgpu | tail -3 > gpurun_out/reference_loop.json
cat gpurun_out/reference_loop.json
