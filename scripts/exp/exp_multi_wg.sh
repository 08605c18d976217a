# the multiplicity kernel with walks per 64-slot chunk (FM_WG = 1: a tighter bound, no three chunks in flight) against the product's
# groups of three — parity first (tests/test_finish_modes.py through the variant), then the times
cd $GRAFT_REPO_ROOT
L=bionumpy_amd/csrc/variants/libbnpk_fm_wg1.so
BNPK_LIB=$L timeout 600 python -m pytest tests/test_finish_modes.py -m gpu -x -q 2>&1 | tail -1
for c in 1 3; do
  for v in product fm_wg1; do
    LL=""; [ "$v" != product ] && LL=$L
    echo "coverage ${c}x $v: $(BNPK_LIB=$LL MB_FINISH_MODE=5 MB_MODE=1 MB_GENOME_LEN=$((7500000000 / c)) timeout 300 python scripts/microbench.py 50000000 2 2>/dev/null | grep 'finish.multi')"
  done
done
echo "random 21-mers fm_wg1: $(BNPK_LIB=$L MB_K=21 timeout 300 python scripts/microbench.py 50000000 2 2>/dev/null | grep 'finish.multi')"
