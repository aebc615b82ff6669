#!/bin/bash
# usage (GPU box): bash tools/tl_env.sh TAG "ENV=..."  -- kernel-trace timeline of one production step under an environment
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-fk --no-kernel-timing --no-phase1"
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_${TAG} -- python3 $R/bench.py $ARGS > $R/gpurun_out/tl_${TAG}.log 2>&1 &&
python3 $R/tools/trace_timeline.py $(ls $R/gpurun_out/tl_${TAG}/*/*kernel_trace.csv | head -1) > $R/gpurun_out/timeline_${TAG}.txt && rm -rf $R/gpurun_out/tl_${TAG}
