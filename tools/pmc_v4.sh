#!/bin/bash
# usage (GPU box): bash tools/pmc_v4.sh "enc3 fwd 1 64 4"   -- SQ counters of one v4 gather launch shape (two passes)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
A=${1:-"enc3 fwd 1 64 4"}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_v4_a -- python3 $R/tools/stamp_v4.py $A > $R/gpurun_out/pmc_v4_a.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmc_v4_b -- python3 $R/tools/stamp_v4.py $A > $R/gpurun_out/pmc_v4_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $R/gpurun_out/pmc_v4_c -- python3 $R/tools/stamp_v4.py $A > $R/gpurun_out/pmc_v4_c.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ("a", "b", "c"):
    files = glob.glob("$R/gpurun_out/pmc_v4_%s/**/*counter_collection.csv" % tag, recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if "gather_gemm4" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("%-34s n=%d  last=%.4g  mean=%.4g" % (k, len(v), v[-1], sum(v) / len(v)))
PY
