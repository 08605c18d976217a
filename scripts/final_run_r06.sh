set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out/r06
mkdir -p $R
timeout 900 python bench.py --steps 5 --warmup 2 > $R/bench_50m_n1.json 2> $R/bench.err; cut -c1-400 $R/bench_50m_n1.json
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed --no-extra"
timeout 300 $B --k 21 > $R/bench_50m_n1_k21.json 2>> $R/bench.err
timeout 300 $B --mode genome --genome-len 7500000000 > $R/bench_50m_n1_genome_1x.json 2>> $R/bench.err
timeout 300 $B --mode genome --genome-len 2500000000 > $R/bench_50m_n1_genome_3x.json 2>> $R/bench.err
timeout 300 $B --mode genome > $R/bench_50m_n1_genome.json 2>> $R/bench.err
timeout 1200 bash scripts/profile_r04.sh "" > $R/profile.log 2>&1
python scripts/summarize_profile.py gpurun_out/prof $R/r06 > $R/summary.log 2>&1; tail -3 $R/summary.log
rm -rf gpurun_out/prof
timeout 1200 bash scripts/profile_r04.sh _k21 --k 21 > $R/profile_k21.log 2>&1
python scripts/summarize_profile.py gpurun_out/prof_k21 $R/r06_k21 > $R/summary_k21.log 2>&1; tail -3 $R/summary_k21.log
rm -rf gpurun_out/prof_k21
timeout 900 bash scripts/profile_sq.sh > $R/sq.log 2>&1; cp gpurun_out/sq/sq_counters.json $R/r06_sq_counters.json 2>/dev/null
rm -rf gpurun_out/sq/pmc1 gpurun_out/sq/pmc2
MB_L1_RING=1 SQ_TAG=_l1_ring timeout 900 bash scripts/profile_sq.sh > $R/sq_ring.log 2>&1; cp gpurun_out/sq/sq_counters_l1_ring.json $R/r06_sq_counters_l1_ring.json 2>/dev/null; tail -12 $R/sq_ring.log
rm -rf gpurun_out/sq/pmc1 gpurun_out/sq/pmc2
timeout 900 python scripts/bench_configs.py 50000000 > $R/configs.json 2> $R/configs.err
timeout 600 python scripts/exp/exp_reference_loop.py 8000000 31 2>/dev/null | tail -1 > $R/reference_loop.json
timeout 300 python bench.py --virtual-ranks 8 --reads 16000000 --steps 1 --warmup 1 --mode genome 2>/dev/null | tail -1 > $R/virtual8_genome.json
timeout 900 bash scripts/exp/exp_finish_rules.sh "1 2 3 4 5 6 8 12 20 30 40 60 100" "0" > $R/coverage_scan.txt 2>&1; tail -3 $R/coverage_scan.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/smoke.log 2>&1; tail -1 $R/smoke.log
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep "passed\|failed\|error" | tail -3 > $R/gpu_tests.txt; cat $R/gpu_tests.txt
du -sh gpurun_out/r06
