# interleaved A/B of the whole graph step under different environments (GPU box): ab_env.sh ROUNDS "ENV_A" "ENV_B" ...
# each arm: python bench.py --steps 100 --warmup 20 without the extras; prints ms/step per arm and round
cd $GRAFT_REPO_ROOT
R=$1; shift
for r in $(seq 1 $R); do
  i=0
  for e in "$@"; do
    v=$(env $e python3 bench.py --steps 100 --warmup 20 --no-extra --no-cpu-baseline --no-fk --no-kernel-timing --no-phase1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f %s' % (d['ms_per_step'], d['config']['tuned']))")
    echo "round $r arm $i [$e]: $v"
    i=$((i+1))
  done
done
