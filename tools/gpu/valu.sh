#!/bin/bash
# vector-ALU counters of the kernels of a chain batch
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && export BFMMM_BATCH_SPLIT=1
rm -rf /tmp/v1 /tmp/v2
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace -d /tmp/v1 -o r -f csv -- python3 $GRAFT_REPO_ROOT/tools/prof_workload.py --workload ${1:-nu_z} --chains 8 --steps 60 > $O/v1.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/sq_counters.py /tmp/v1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/v2 -o r -f csv -- python3 $GRAFT_REPO_ROOT/tools/prof_workload.py --workload ${1:-nu_z} --chains 8 --steps 60 > $O/v2.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/sq_counters.py /tmp/v2
