# where the multiplicity kernel's time goes on low-coverage reads: ablation builds (timing only: their results are wrong)
#   scripts/exp/build_variant.sh fm_abl1 finish_multi.hip -DFM_ABL=1   (1 no look-back, 4 no output stores, 16 a ring that fits the L2, 32 walks of one step)
cd $GRAFT_REPO_ROOT
for c in ${COVS:-1 3}; do
  for v in ${VARIANTS:-product fm_abl1 fm_abl17 fm_abl5 fm_abl21 fm_abl32 fm_abl37}; do
    L=""; [ "$v" != product ] && L=bionumpy_amd/csrc/variants/libbnpk_$v.so
    echo "coverage ${c}x $v: $(BNPK_LIB=$L MB_FINISH_MODE=5 MB_MODE=1 MB_GENOME_LEN=$((7500000000 / c)) timeout 300 python scripts/microbench.py 50000000 2 2>/dev/null | grep 'finish.multi')"
  done
done
