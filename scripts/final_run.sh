set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out/r02
mkdir -p $R
timeout 1500 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1; tail -2 $R/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 > $R/bench_50m_n1.json 2> $R/bench.err; cat $R/bench_50m_n1.json | cut -c1-600
timeout 1500 bash scripts/profile_r02.sh > $R/profile.log 2>&1
python scripts/summarize_profile.py gpurun_out/prof $R/r02 > $R/summary.log 2>&1; tail -3 $R/summary.log
python scripts/summarize_hostfed.py gpurun_out/prof/hostfed $R/r02_hostfed_trace.txt; cp gpurun_out/prof/bench_hostfed.json $R/ 2>/dev/null
ls gpurun_out/prof/hostfed | head
timeout 400 python bench.py --steps 3 --warmup 1 --mode genome --no-cpu-baseline --no-host-fed > $R/bench_genome.json 2> $R/bench_genome.err; cat $R/bench_genome.json | cut -c1-400
timeout 900 bash scripts/profile_sq.sh > $R/sq.log 2>&1; cp gpurun_out/sq/r02_sq_counters.json $R/ 2>/dev/null; tail -5 $R/sq.log
timeout 900 python scripts/bench_configs.py 50000000 > $R/configs.json 2> $R/configs.err; cat $R/configs.json | cut -c1-1500
timeout 300 python bench.py --virtual-ranks 8 --reads 16000000 --steps 1 --warmup 1 > $R/virtual8.json 2>&1; tail -1 $R/virtual8.json | cut -c1-600
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/smoke.log 2>&1; tail -1 $R/smoke.log
# only the summaries travel back (gpurun merges at most 64 MiB)
rm -rf gpurun_out/prof gpurun_out/sq/pmc1 gpurun_out/sq/pmc2
du -sh gpurun_out
