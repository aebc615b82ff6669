#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh TAG
# A plain bench run first (fills the autotune cache so that the profiled runs launch only the timed kernels), then
# profiler passes of the same command: kernel trace + stats of the production graph, the same with one stream and
# eager launches (per-kernel times without two streams sharing the CUs), FETCH_SIZE, WRITE_SIZE (separate counter passes,
# no other trace domains).  Outputs under gpurun_out/prof_TAG_*; tools/pmc_traffic.py turns them into
# profiles/TAG_traffic.json and profiles/TAG_kernel_stats.csv.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-fk --no-kernel-timing --no-phase1 --no-extra"
export HMVAE_TUNE_CACHE=$R/gpurun_out/tune_${TAG}.json
rm -f $HMVAE_TUNE_CACHE
python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_${TAG}_bench.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_stats.log 2>&1 &&
HMVAE_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_serial -- python3 $R/bench.py $ARGS --graph 0 > $R/gpurun_out/prof_${TAG}_serial.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py $ARGS --graph 0 > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $R/bench.py $ARGS --graph 0 > $R/gpurun_out/prof_${TAG}_write.log 2>&1 &&
echo profile done $TAG
