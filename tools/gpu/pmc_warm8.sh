#!/bin/bash
# SQ / MFMA counters of the 8-chain warm batch on one stream
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BFMMM_BATCH_SPLIT=1
for set in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pm1
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pm1 -o w -f csv -- python3 $R/tools/prof_workload.py --workload warm --chains ${1:-8} --steps 60 > /dev/null 2>&1
  python3 $R/tools/sq_counters.py /tmp/pm1
done
