#!/bin/bash
# usage: tools/pmc.sh TAG -- python3 tools/mb_layer.py ...   (three counter passes, csv under gpurun_out/pmc_TAG_k)
TAG=$1; shift; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_${TAG}_1 -- "$@" > $R/gpurun_out/pmc_${TAG}_1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pmc_${TAG}_2 -- "$@" > $R/gpurun_out/pmc_${TAG}_2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${TAG}_3 -- "$@" > $R/gpurun_out/pmc_${TAG}_3.log 2>&1
echo pmc done $TAG
