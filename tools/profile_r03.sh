#!/bin/bash
# round-3 evidence in one GPU call: the step (graph + serial + traffic passes), the FK kernels, the two extra bench sections,
# counter passes over the first decoder layer's weight gradient (48 x 64 tiles as shipped, wide tiles)
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
bash $R/tools/profile_round.sh $TAG &&
bash $R/tools/profile_fk.sh $TAG &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_sampling -- python3 $R/bench.py --only sampling --steps 10 > $R/gpurun_out/prof_${TAG}_sampling.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_traj -- python3 $R/bench.py --only traj --steps 20 --warmup 5 > $R/gpurun_out/prof_${TAG}_traj.log 2>&1 &&
bash $R/tools/pmc_wgrad3.sh dec0 2 1 &&
bash $R/tools/pmc_wgrad3.sh dec0 4 1 &&
bash $R/tools/pmc_wgrad3.sh dec0 pq 1 &&
echo all profiles done
