#!/bin/bash
# usage (on the GPU box): bash tools/profile_fk.sh TAG
# rocprofv3 passes over the config-4 FK kernels (tools/fk_profile.py = bench.py's fk_roofline leg alone): kernel trace +
# stats, then FETCH_SIZE and WRITE_SIZE in separate counter passes.  tools/pmc_traffic.py TAG fk turns them into
# profiles/TAG_fk_traffic.json and profiles/TAG_fk_kernel_stats.csv.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_fk_stats -- python3 $R/tools/fk_profile.py > $R/gpurun_out/prof_${TAG}_fk_stats.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fk_fetch -- python3 $R/tools/fk_profile.py > $R/gpurun_out/prof_${TAG}_fk_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fk_write -- python3 $R/tools/fk_profile.py > $R/gpurun_out/prof_${TAG}_fk_write.log 2>&1 &&
echo fk profile done $TAG
