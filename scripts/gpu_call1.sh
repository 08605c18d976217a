set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05c1
mkdir -p $R
timeout 700 python -m pytest tests -m gpu -x -q > $R/pytest.log 2>&1; tail -5 $R/pytest.log
for v in rc_u4 rc_u4_pad rc_u4_o1 rc_u2; do
  timeout 120 python scripts/bin/$v/scripts/exp/rc_repro.py quiet > $R/$v.log 2>&1; tail -12 $R/$v.log
done
timeout 120 python scripts/bin/rc_u4/scripts/exp/rc_repro.py poison > $R/rc_u4_poison.log 2>&1; tail -12 $R/rc_u4_poison.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-host-fed --cpu-sample-reads 50000 > $R/bench.json 2> $R/bench.err; cut -c1-600 $R/bench.json
timeout 300 python bench.py --from-file /dev/shm/bnpk_ff.fq --reads 8000000 --steps 2 --warmup 1 > $R/from_file.json 2> $R/from_file.err; cat $R/from_file.json; tail -3 $R/from_file.err
rm -f /dev/shm/bnpk_ff.fq
