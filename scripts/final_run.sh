set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01b
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r01b/pytest_gpu.log 2>&1; tail -2 gpurun_out/r01b/pytest_gpu.log
timeout 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r01b/bench_50m_n1.json 2> gpurun_out/r01b/bench.err; cat gpurun_out/r01b/bench_50m_n1.json
rm -rf gpurun_out/prof
timeout 900 bash scripts/profile_r01.sh > gpurun_out/r01b/profile.log 2>&1
python scripts/summarize_profile.py gpurun_out/prof gpurun_out/r01b/r01 > gpurun_out/r01b/summary.log 2>&1; tail -3 gpurun_out/r01b/summary.log
timeout 300 python bench.py --steps 3 --warmup 1 --mode genome --no-cpu-baseline > gpurun_out/r01b/bench_genome.json 2> gpurun_out/r01b/bench_genome.err; cat gpurun_out/r01b/bench_genome.json
timeout 600 python scripts/bench_configs.py 50000000 > gpurun_out/r01b/configs.json 2> gpurun_out/r01b/configs.err; cat gpurun_out/r01b/configs.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r01b/smoke.log 2>&1; tail -1 gpurun_out/r01b/smoke.log
