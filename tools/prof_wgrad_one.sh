set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfgs in "dec0 4 1" "enc3 4 1" "dec0 2 1"; do
  set -- $cfgs
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$1_$2 -o p -- python3 tools/wgrad_one.py $1 $2 $3 both 10 > gpurun_out/prof_$1_$2.log 2>&1
done
