set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05
mkdir -p $R
timeout 900 python bench.py --steps 5 --warmup 2 > $R/bench_50m_n1.json 2> $R/bench.err; cut -c1-300 $R/bench_50m_n1.json
timeout 1200 bash scripts/profile_r04.sh "" > $R/profile.log 2>&1
python scripts/summarize_profile.py gpurun_out/prof $R/r05 > $R/summary.log 2>&1; tail -3 $R/summary.log
rm -rf gpurun_out/prof
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed --no-extra"
timeout 300 $B --mode genome > $R/bench_50m_n1_genome.json 2>> $R/bench.err
timeout 300 $B --k 21 > $R/bench_50m_n1_k21.json 2>> $R/bench.err
du -sh $R
