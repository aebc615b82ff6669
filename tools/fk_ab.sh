cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for e in "X=0" "HMVAE_TF_ALIAS=0"; do
    echo "== round $r [$e]"; env $e python3 tools/fk_bench.py 2>/dev/null | grep "decode_tail_fwd"
  done
done
