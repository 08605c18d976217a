set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05
mkdir -p $R
timeout 1200 bash scripts/profile_r04.sh "" > $R/profile.log 2>&1
python scripts/summarize_profile.py gpurun_out/prof $R/r05 > $R/summary.log 2>&1; head -12 $R/r05_kernel_stats.txt | cut -c1-120
rm -rf gpurun_out/prof
cp $R/r05_pmc.json profiles/r05_pmc.json
timeout 900 python bench.py --steps 5 --warmup 2 > $R/bench_50m_n1.json 2> $R/bench.err; cut -c1-700 $R/bench_50m_n1.json
