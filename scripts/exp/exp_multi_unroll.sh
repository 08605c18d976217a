# the multiplicity kernel's walk unrolling (FM_UNROLL: 4 in the product) now that the walks are bounded per group
cd $GRAFT_REPO_ROOT
for c in 1 3; do
  for v in product fm_u1 fm_u2 fm_u8; do
    L=""; [ "$v" != product ] && L=bionumpy_amd/csrc/variants/libbnpk_$v.so
    echo "coverage ${c}x $v: $(BNPK_LIB=$L MB_FINISH_MODE=5 MB_MODE=1 MB_GENOME_LEN=$((7500000000 / c)) timeout 300 python scripts/microbench.py 50000000 2 2>/dev/null | grep 'finish.multi')"
  done
done
for v in product fm_u1 fm_u2 fm_u8; do
  L=""; [ "$v" != product ] && L=bionumpy_amd/csrc/variants/libbnpk_$v.so
  echo "random 21-mers $v: $(BNPK_LIB=$L MB_K=21 timeout 300 python scripts/microbench.py 50000000 2 2>/dev/null | grep 'finish.multi')"
done
