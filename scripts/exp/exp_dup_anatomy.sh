# per-kernel times of the workgroup-table path on reads at 5x / 12x coverage (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
for c in ${COVS:-5 12}; do
  rm -rf /tmp/prof_dup
  MB_MODE=1 MB_GENOME_LEN=$((7500000000 / c)) rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dup -o x -- python $GRAFT_REPO_ROOT/scripts/microbench.py 50000000 2 > /tmp/prof_dup.log 2>&1
  echo "== coverage ${c}x"
  python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/prof_dup/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:12]:
    print("%-70s calls %5s  avg %9.3f ms  total %9.3f ms" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
PY
done
