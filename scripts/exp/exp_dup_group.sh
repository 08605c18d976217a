# the workgroup table's inserts in flight per lane (FD_GROUP_N: 4 in the product) on reads at 5x / 12x / 30x coverage
cd $GRAFT_REPO_ROOT
for c in 5 12 30; do
  for v in product fd_g2 fd_g8 fd_g16; do
    L=""; [ "$v" != product ] && L=bionumpy_amd/csrc/variants/libbnpk_$v.so
    echo "coverage ${c}x $v: $(BNPK_LIB=$L MB_MODE=1 MB_GENOME_LEN=$((7500000000 / c)) timeout 300 python scripts/microbench.py 50000000 2 2>/dev/null | grep 'finish.dup')"
  done
done
