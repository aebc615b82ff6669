#!/bin/bash
# usage (GPU box): bash tools/pmc_gather.sh dec3 f  -- counter passes over one layer's gather-GEMM
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
L=${1:-dec3}; S=${2:-f}
export HMVAE_TUNE_CACHE=$R/gpurun_out/tune_pmcg.json
python3 $R/tools/gather_only.py $L $S > $R/gpurun_out/pmcg_warm.log 2>&1
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcg_${L}${S}_$tag -- python3 $R/tools/gather_only.py $L $S > $R/gpurun_out/pmcg_$tag.log 2>&1 || echo "pass failed: $C"
done
python3 $R/tools/pmc_wgrad_summary.py ${L}${S} pmcg gather
echo pmc done
