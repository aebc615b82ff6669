#!/bin/bash
# usage (GPU box): bash tools/pmc_wgrad3.sh LAYER MODE SPLIT  -- counter passes (each its own run, kernel trace only beside
# it) over one layer's weight-gradient launch in one configuration (tools/wgrad_one.py); summary by tools/pmc_wgrad_summary.py
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
L=${1:-dec0}; M=${2:-2}; S=${3:-1}
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcw_${L}m${M}s${S}_$tag -- python3 $R/tools/wgrad_one.py $L $M $S fused 12 > $R/gpurun_out/pmcw3_$tag.log 2>&1 || echo "pass failed: $C"
done
python3 $R/tools/pmc_wgrad_summary.py ${L}m${M}s${S} > $R/gpurun_out/pmc_wgrad_${L}_m${M}_s${S}.txt
echo pmc done $L $M $S
