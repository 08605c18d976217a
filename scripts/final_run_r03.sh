set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out/r03
mkdir -p $R
timeout 1800 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1; tail -2 $R/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 > $R/bench_50m_n1.json 2> $R/bench.err; cat $R/bench_50m_n1.json | cut -c1-600
timeout 600 python bench.py --steps 3 --warmup 1 --mode genome --no-host-fed > $R/bench_50m_n1_genome.json 2> $R/bench_genome.err; cat $R/bench_50m_n1_genome.json | cut -c1-400
timeout 1500 bash scripts/profile_r03.sh 50000000 uniform > $R/profile.log 2>&1
python scripts/summarize_profile.py gpurun_out/prof $R/r03 > $R/summary.log 2>&1; tail -3 $R/summary.log
python scripts/summarize_hostfed.py gpurun_out/prof/hostfed $R/r03_hostfed_trace.txt; cp gpurun_out/prof/bench_hostfed.json $R/ 2>/dev/null
rm -rf gpurun_out/prof
timeout 1200 bash scripts/profile_r03.sh 50000000 genome _genome > $R/profile_genome.log 2>&1
python scripts/summarize_profile.py gpurun_out/prof_genome $R/r03_genome > $R/summary_genome.log 2>&1; tail -3 $R/summary_genome.log
rm -rf gpurun_out/prof_genome
timeout 900 bash scripts/profile_sq.sh > $R/sq.log 2>&1; cp gpurun_out/sq/r03_sq_counters.json $R/ 2>/dev/null; tail -5 $R/sq.log
rm -rf gpurun_out/sq/pmc1 gpurun_out/sq/pmc2
MB_MODE=1 SQ_TAG=_genome timeout 900 bash scripts/profile_sq.sh > $R/sq_genome.log 2>&1; cp gpurun_out/sq/r03_sq_counters_genome.json $R/ 2>/dev/null; tail -5 $R/sq_genome.log
rm -rf gpurun_out/sq/pmc1 gpurun_out/sq/pmc2
timeout 900 python scripts/bench_configs.py 50000000 > $R/configs.json 2> $R/configs.err; cat $R/configs.json | cut -c1-1500
timeout 300 python bench.py --virtual-ranks 8 --reads 16000000 --steps 1 --warmup 1 > $R/virtual8.json 2>&1; tail -1 $R/virtual8.json | cut -c1-600
timeout 300 python bench.py --virtual-ranks 8 --reads 16000000 --steps 1 --warmup 1 --mode genome > $R/virtual8_genome.json 2>&1; tail -1 $R/virtual8_genome.json | cut -c1-600
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/smoke.log 2>&1; tail -1 $R/smoke.log
du -sh gpurun_out
