#!/bin/bash
# usage (on the GPU box): bash tools/profile_step.sh TAG   -- kernel trace + stats of the production graph step only
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-fk --no-kernel-timing --no-phase1"
export HMVAE_TUNE_CACHE=$R/gpurun_out/tune_${TAG}.json
rm -f $HMVAE_TUNE_CACHE
python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_${TAG}_bench.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_stats.log 2>&1 &&
cp $(ls $R/gpurun_out/prof_${TAG}_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/prof_${TAG}_kernel_stats.csv && echo profile done $TAG
