#!/bin/bash
# kernel-trace timelines of the graph step under two environments (GPU box): trace_ab.sh TAG_A "ENV_A" TAG_B "ENV_B"
# (rocprofv3 needs the program right after "--": the environment is exported in this shell instead)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  TAG=$1; ENVS=$2; shift 2
  ( for kv in $ENVS; do export "$kv"; done
    rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -- python3 $R/bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-fk --no-kernel-timing --no-phase1 > $R/gpurun_out/trace_$TAG.log 2>&1 ) || exit 1
  f=$(find $R/gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_timeline.py $f 3 > $R/gpurun_out/timeline_$TAG.txt
  rm -rf $R/gpurun_out/trace_$TAG
done
