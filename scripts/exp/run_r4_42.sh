cd $GRAFT_REPO_ROOT
for v in abl1 abl2; do echo $v; (cd scripts/bin/$v && timeout 600 python scripts/microbench.py 50000000 2 2>&1 | grep "fastq_") ; done
