cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for e in "X=0" "HMVAE_FK_BWD_OCC=2" "HMVAE_TAIL3_FR=32"; do
    echo "== round $r [$e]"; env $e python3 tools/fk_bench.py 2>/dev/null | grep "fk_bwd\|tail_loss3"
  done
done
